// Issue-rate microbenchmark for gfx950: cycles per instruction per wave for VALU / packed VALU / SALU
// streams at 1, 2 and 4 waves per SIMD (development aid, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 2000

#define KERNEL(name, body)                                                                        \
__global__ void name(unsigned long long *out, float seed, unsigned us) {                          \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    float b0 = 0.5f, b1 = 1.5f;                                                                  \
    unsigned s0 = us, s1 = us + 3, s2 = us + 5, s3 = us + 7;                                      \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
    for (int i = 0; i < ITER; ++i) {                                                              \
        asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                     "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b0), "v"(b1) : "scc", "vcc");    \
    }                                                                                             \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;       \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3) == 12345.678f) out[0] = 0; \
}

// 16 instructions per asm body (8 independent registers, used twice)
KERNEL(k_fma,  REP16("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n"))
KERNEL(k_add,  REP16("v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n v_add_f32 %2, %2, %12\n v_add_f32 %3, %3, %12\n v_add_f32 %4, %4, %12\n v_add_f32 %5, %5, %12\n v_add_f32 %6, %6, %12\n v_add_f32 %7, %7, %12\n"))
KERNEL(k_max,  REP16("v_max_f32 %0, %0, %12\n v_max_f32 %1, %1, %12\n v_max_f32 %2, %2, %12\n v_max_f32 %3, %3, %12\n v_max_f32 %4, %4, %12\n v_max_f32 %5, %5, %12\n v_max_f32 %6, %6, %12\n v_max_f32 %7, %7, %12\n"))
KERNEL(k_floor,REP16("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n"))
KERNEL(k_addu, REP16("v_add_u32 %0, %0, %12\n v_add_u32 %1, %1, %12\n v_add_u32 %2, %2, %12\n v_add_u32 %3, %3, %12\n v_add_u32 %4, %4, %12\n v_add_u32 %5, %5, %12\n v_add_u32 %6, %6, %12\n v_add_u32 %7, %7, %12\n"))
KERNEL(k_mov,  REP16("v_mov_b32 %0, %12\n v_mov_b32 %1, %12\n v_mov_b32 %2, %12\n v_mov_b32 %3, %12\n v_mov_b32 %4, %12\n v_mov_b32 %5, %12\n v_mov_b32 %6, %12\n v_mov_b32 %7, %12\n"))
KERNEL(k_salu, REP16("s_and_b32 %8, %8, %9\n s_lshl_b32 %9, %9, 1\n s_and_b32 %10, %10, %11\n s_add_u32 %11, %11, 3\n s_and_b32 %8, %8, %9\n s_lshl_b32 %9, %9, 1\n s_and_b32 %10, %10, %11\n s_add_u32 %11, %11, 3\n"))
KERNEL(k_mix,  REP16("v_fma_f32 %0, %0, %12, %13\n s_and_b32 %8, %8, %9\n v_fma_f32 %1, %1, %12, %13\n s_lshl_b32 %9, %9, 1\n v_fma_f32 %2, %2, %12, %13\n s_and_b32 %10, %10, %11\n v_fma_f32 %3, %3, %12, %13\n s_add_u32 %11, %11, 3\n"))


#define V8(op, tail) REP16(op " %0, %0" tail "\n " op " %1, %1" tail "\n " op " %2, %2" tail "\n " op " %3, %3" tail "\n " op " %4, %4" tail "\n " op " %5, %5" tail "\n " op " %6, %6" tail "\n " op " %7, %7" tail "\n")
KERNEL(k_sub,   V8("v_sub_f32", ", %12"))
KERNEL(k_mul,   V8("v_mul_f32", ", %12"))
KERNEL(k_min,   V8("v_min_f32", ", %12"))
KERNEL(k_trunc, V8("v_trunc_f32", ""))
KERNEL(k_fract, V8("v_fract_f32", ""))
KERNEL(k_rndne, V8("v_rndne_f32", ""))
KERNEL(k_cvtu,  V8("v_cvt_u32_f32", ""))
KERNEL(k_cvtf,  V8("v_cvt_f32_u32", ""))
KERNEL(k_med3,  V8("v_med3_f32", ", %12, %13"))
KERNEL(k_max3,  V8("v_max3_f32", ", %12, %13"))
KERNEL(k_addcl, V8("v_add_f32_e64", ", %12 clamp"))
KERNEL(k_cmp,   REP16("v_cmp_gt_f32 vcc, %0, %12\n v_cmp_gt_f32 vcc, %1, %12\n v_cmp_gt_f32 vcc, %2, %12\n v_cmp_gt_f32 vcc, %3, %12\n v_cmp_gt_f32 vcc, %4, %12\n v_cmp_gt_f32 vcc, %5, %12\n v_cmp_gt_f32 vcc, %6, %12\n v_cmp_gt_f32 vcc, %7, %12\n"))
KERNEL(k_cnd,   REP16("v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %1, %1, %12, vcc\n v_cndmask_b32 %2, %2, %12, vcc\n v_cndmask_b32 %3, %3, %12, vcc\n v_cndmask_b32 %4, %4, %12, vcc\n v_cndmask_b32 %5, %5, %12, vcc\n v_cndmask_b32 %6, %6, %12, vcc\n v_cndmask_b32 %7, %7, %12, vcc\n"))
KERNEL(k_and,   V8("v_and_b32", ", %12"))
KERNEL(k_maxi,  V8("v_max_i32", ", %12"))
KERNEL(k_or,    V8("v_or_b32", ", %12"))
KERNEL(k_or3,   V8("v_or3_b32", ", %12, %13"))
KERNEL(k_bfi,   V8("v_bfi_b32", ", %12, %13"))
KERNEL(k_andor, V8("v_and_or_b32", ", %12, %13"))
KERNEL(k_lshladd, V8("v_lshl_add_u32", ", 3, %13"))
KERNEL(k_addco, REP16("v_add_co_u32 %0, vcc, %0, %12\n v_add_co_u32 %1, vcc, %1, %12\n v_add_co_u32 %2, vcc, %2, %12\n v_add_co_u32 %3, vcc, %3, %12\n v_add_co_u32 %4, vcc, %4, %12\n v_add_co_u32 %5, vcc, %5, %12\n v_add_co_u32 %6, vcc, %6, %12\n v_add_co_u32 %7, vcc, %7, %12\n"))
KERNEL(k_fmac,  V8("v_fmac_f32", ", %12"))
KERNEL(k_cmpu,  REP16("v_cmp_le_u32 vcc, %12, %0\n v_cmp_le_u32 vcc, %12, %1\n v_cmp_le_u32 vcc, %12, %2\n v_cmp_le_u32 vcc, %12, %3\n v_cmp_le_u32 vcc, %12, %4\n v_cmp_le_u32 vcc, %12, %5\n v_cmp_le_u32 vcc, %12, %6\n v_cmp_le_u32 vcc, %12, %7\n"))
KERNEL(k_fmadpp, V8("v_fmac_f32_dpp", ", %12 row_newbcast:3 row_mask:0xf bank_mask:0xf"))
KERNEL(k_fmas,  V8("v_fma_f32", ", s9, %13"))
KERNEL(k_fmasabs, REP16("v_fma_f32 %0, |s9|, %12, %0\n v_fma_f32 %1, |s9|, %12, %1\n v_fma_f32 %2, |s9|, %12, %2\n v_fma_f32 %3, |s9|, %12, %3\n v_fma_f32 %4, |s9|, %12, %4\n v_fma_f32 %5, |s9|, %12, %5\n v_fma_f32 %6, |s9|, %12, %6\n v_fma_f32 %7, |s9|, %12, %7\n"))
KERNEL(k_fmaacc, REP16("v_fma_f32 %0, %13, %12, %0\n v_fma_f32 %1, %13, %12, %1\n v_fma_f32 %2, %13, %12, %2\n v_fma_f32 %3, %13, %12, %3\n v_fma_f32 %4, %13, %12, %4\n v_fma_f32 %5, %13, %12, %5\n v_fma_f32 %6, %13, %12, %6\n v_fma_f32 %7, %13, %12, %7\n"))
KERNEL(k_fmaacc2, REP16("v_fma_f32 %0, %13, %12, %0\n v_fma_f32 %1, %13, %12, %1\n v_fma_f32 %0, %13, %12, %0\n v_fma_f32 %1, %13, %12, %1\n v_fma_f32 %0, %13, %12, %0\n v_fma_f32 %1, %13, %12, %1\n v_fma_f32 %0, %13, %12, %0\n v_fma_f32 %1, %13, %12, %1\n"))
KERNEL(k_addus, V8("v_add_u32", ", s9"))


KERNEL(k_salu2, REP16("s_and_b32 %8, %8, %9\n s_and_b32 %9, %9, %10\n s_and_b32 %10, %10, %11\n s_and_b32 %11, %11, %8\n s_and_b32 %8, %8, %9\n s_and_b32 %9, %9, %10\n s_and_b32 %10, %10, %11\n s_and_b32 %11, %11, %8\n"))
KERNEL(k_mix2, REP16("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n s_and_b32 %8, %8, %9\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n s_lshl_b32 %9, %9, 1\n v_fma_f32 %4, %4, %12, %13\n s_and_b32 %10, %10, %11\n"))
KERNEL(k_mix3, REP16("v_fma_f32 %0, %0, %12, %13\n s_and_b32 %8, %8, %9\n s_lshl_b32 %9, %9, 1\n v_fma_f32 %1, %1, %12, %13\n s_and_b32 %10, %10, %11\n s_add_u32 %11, %11, 3\n v_fma_f32 %2, %2, %12, %13\n s_and_b32 %8, %8, %10\n"))

// packed: operate on register pairs (a0,a1),(a2,a3),(a4,a5),(a6,a7)
#define KERNEL_PK(name, body)                                                                     \
__global__ void name(unsigned long long *out, float seed, unsigned us) {                          \
    typedef float v2f __attribute__((ext_vector_type(2)));                                        \
    v2f a0 = {seed, seed + 1}, a1 = {seed + 2, seed + 3}, a2 = {seed + 4, seed + 5}, a3 = {seed + 6, seed + 7}; \
    v2f b0 = {0.5f, 0.25f}, b1 = {1.5f, 2.5f};                                                   \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
    for (int i = 0; i < ITER; ++i) {                                                              \
        asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));           \
    }                                                                                             \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;       \
    if (a0.x + a1.x + a2.y + a3.y == 12345.678f) out[0] = 0;                                      \
}
// 8 instructions per asm body
KERNEL_PK(k_pkfma, REP16("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"))
KERNEL_PK(k_pkadd, REP16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"))
KERNEL_PK(k_pkmul, REP16("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"))

template <typename K>
void run(const char *name, K kern, int instr_per_body) {
    unsigned long long *d; hipMalloc(&d, 256 * 16 * 8);
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, 1.0f, 1u);
        { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return; } }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, 1.0f, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= h.size();
        double n = (double)ITER * instr_per_body * 16;
        // s_memtime ticks at 100 MHz-derived constant rate? report both tick-based and wall-based
        printf("%-8s threads/CU %4d  waves/SIMD %.2f  ticks/instr/wave %.3f   wall ns/instr/wave %.3f  -> instr/ns/SIMD %.3f\n",
               name, threads, threads / 256.0, avg / n, ms * 1e6 / n, n * (threads / 64) / 4.0 / (ms * 1e6));
    }
    hipFree(d);
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("start\n");
    std::string w = argc > 1 ? argv[1] : "all";
#define RUN(n, k, c) if (w == "all" || w == n) run(n, k, c);
    RUN("fma", k_fma, 8) RUN("add", k_add, 8) RUN("max", k_max, 8) RUN("floor", k_floor, 8)
    RUN("add_u32", k_addu, 8) RUN("mov", k_mov, 8) RUN("salu", k_salu, 8) RUN("mix", k_mix, 8)
    RUN("salu2", k_salu2, 8) RUN("mix2", k_mix2, 8) RUN("mix3", k_mix3, 8) RUN("pk_fma", k_pkfma, 4) RUN("pk_add", k_pkadd, 4) RUN("pk_mul", k_pkmul, 4)
    RUN("sub", k_sub, 8) RUN("mul", k_mul, 8) RUN("min", k_min, 8) RUN("trunc", k_trunc, 8) RUN("fract", k_fract, 8)
    RUN("rndne", k_rndne, 8) RUN("cvt_u32_f32", k_cvtu, 8) RUN("cvt_f32_u32", k_cvtf, 8) RUN("med3", k_med3, 8)
    RUN("max3", k_max3, 8) RUN("add_clamp", k_addcl, 8) RUN("cmp", k_cmp, 8) RUN("cndmask", k_cnd, 8)
    RUN("and", k_and, 8) RUN("max_i32", k_maxi, 8)
    RUN("or", k_or, 8) RUN("or3", k_or3, 8) RUN("bfi", k_bfi, 8) RUN("and_or", k_andor, 8) RUN("lshl_add", k_lshladd, 8)
    RUN("add_co", k_addco, 8) RUN("fmac", k_fmac, 8) RUN("cmp_u32", k_cmpu, 8) RUN("fmac_dpp", k_fmadpp, 8)
    RUN("fma_s", k_fmas, 8) RUN("fma_sabs", k_fmasabs, 8) RUN("fma_acc", k_fmaacc, 8) RUN("fma_acc2", k_fmaacc2, 8) RUN("addu_s", k_addus, 8)
    return 0;
}

// Slot-cost microbenchmark for the event loop of k_output_counts (development aid, not part of the product):
// cycles per "event" for candidate instruction mixes at 1, 2, 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define ITER 1000

// operands: %0..%3 acc/weight-ish VGPRs (rw), %4 addr VGPR tmp (rw), %5..%6 w pair (rw 64-bit), %7 sgpr (rw)
//           %8 lane offset VGPR, %9 counts VGPR, %10 sgpr multiplier, %11 offsets VGPR
#define KERNEL(name, body)                                                                        \
__global__ void __launch_bounds__(1024) name(unsigned long long *out, float seed, unsigned us) {  \
    __shared__ float tile[8192];                                                                  \
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tile[i] = seed + i;                      \
    __syncthreads();                                                                              \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;                                 \
    unsigned addr = 0; double w = 0; unsigned s0 = us;                                            \
    unsigned lane8 = (threadIdx.x & 63) * 8; float cntv = seed * (threadIdx.x & 15);              \
    unsigned offs = (threadIdx.x & 15) * 512; float smul = seed;                                  \
    typedef float v2f __attribute__((ext_vector_type(2)));                                        \
    v2f accp = {seed, seed}, wp = {seed * 0.001f, seed * 0.002f}, half = {0.5f, 0.5f}, two = {2.0f, 2.0f}, mone = {-1.0f, -1.0f}; \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
    for (int i = 0; i < ITER; ++i) {                                                              \
        asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(addr), "+v"(w), "+s"(s0) \
                     : "v"(lane8), "v"(cntv), "s"(smul), "v"(offs), [accp] "v"(accp), [wp] "v"(wp), [half] "v"(half), [two] "v"(two), [mone] "v"(mone) : "scc", "vcc", "s20", "s21"); \
    }                                                                                             \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;       \
    if (a0 + a1 + a2 + a3 + (float)w + (float)s0 + (float)addr + accp.x + accp.y + wp.x == 12345.678f) out[0] = 0; \
}

#define KERNEL4(name, body)                                                                        \
__global__ void __launch_bounds__(1024) name(unsigned long long *out, float seed, unsigned us) {  \
    __shared__ float tile[8192];                                                                  \
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tile[i] = seed + i;                      \
    __syncthreads();                                                                              \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;                                 \
    unsigned addr = 0; double w = 0; unsigned s0 = us; unsigned b2 = 0; float w3 = seed * 0.003f, c3 = seed, v3 = seed * 0.1f;                                            \
    unsigned lane8 = (threadIdx.x & 63) * 8; float cntv = seed * (threadIdx.x & 15);              \
    unsigned offs = 512 + 0 * threadIdx.x; float smul = seed;                                  \
    typedef float v2f __attribute__((ext_vector_type(2)));                                        \
    typedef float v3f __attribute__((ext_vector_type(3))); v3f w96 = {seed, seed, seed}; unsigned lane12 = (threadIdx.x & 63) * 12; unsigned lane4 = (threadIdx.x & 63) * 4; typedef float v4f __attribute__((ext_vector_type(4))); v4f accp4 = {seed, seed, seed, seed}; v2f accp = {seed, seed}, wp = {seed * 0.001f, seed * 0.002f}, half = {0.5f, 0.5f}, two = {2.0f, 2.0f}, mone = {-1.0f, -1.0f}; \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
    for (int i = 0; i < ITER; ++i) {                                                              \
        asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(addr), "+v"(w), "+s"(s0), [b2] "+v"(b2), [w3] "+v"(w3), [c3] "+v"(c3), [v3] "+v"(v3), [w96] "+v"(w96) \
                     : "v"(lane8), "v"(cntv), "s"(smul), "v"(offs), [accp] "v"(accp), [accp4] "v"(accp4), [lane12] "v"(lane12), [lane4] "v"(lane4), [wp] "v"(wp), [half] "v"(half), [two] "v"(two), [mone] "v"(mone) : "scc", "vcc", "s20", "s21", "m0"); \
    }                                                                                             \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;       \
    if (a0 + a1 + a2 + a3 + (float)w + (float)s0 + (float)addr + (float)b2 + w3 + c3 + v3 + w96.x + w96.z + accp.x + accp.y + wp.x == 12345.678f) out[0] = 0; \
}

// current kernel's event: SGPR-sourced offset and count
#define EV_CUR "v_add_u32 %4, %6, %7\n ds_read_b64 %5, %4\n s_waitcnt lgkmcnt(3)\n v_fma_f32 %0, |%9|, %2, %0\n s_cmp_gt_i32 %6, -1\n v_fma_f32 %1, |%9|, %3, %1\n s_cbranch_scc0 1f\n 1:\n"
KERNEL(k_cur, REP8(EV_CUR))
// proposed: DPP row broadcast of VGPR-held offsets/counts
#define EV_DPP "v_add_u32_dpp %4, %10, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n ds_read_b64 %5, %4\n s_waitcnt lgkmcnt(3)\n v_fmac_f32_dpp %0, |%8|, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n s_bitcmp1_b32 %6, 7\n v_fmac_f32_dpp %1, |%8|, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n s_cbranch_scc1 1f\n 1:\n"
KERNEL(k_dpp, REP8(EV_DPP))
// DPP without the LDS read
#define EV_DPPNL "v_add_u32_dpp %4, %10, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %0, |%8|, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n s_bitcmp1_b32 %6, 7\n v_fmac_f32_dpp %1, |%8|, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n s_cbranch_scc1 1f\n 1:\n"
KERNEL(k_dppnl, REP8(EV_DPPNL))
// VALU only (3 per event)
#define EV_V3 "v_add_u32 %4, %6, %7\n v_fma_f32 %0, |%9|, %2, %0\n v_fma_f32 %1, |%9|, %3, %1\n"
KERNEL(k_v3, REP8(EV_V3))
// VALU + LDS, no scalar
#define EV_V3L "v_add_u32 %4, %6, %7\n ds_read_b64 %5, %4\n s_waitcnt lgkmcnt(3)\n v_fma_f32 %0, |%9|, %2, %0\n v_fma_f32 %1, |%9|, %3, %1\n"
KERNEL(k_v3l, REP8(EV_V3L))
// VALU + one scalar compare, no branch
#define EV_V3S "v_add_u32 %4, %6, %7\n v_fma_f32 %0, |%9|, %2, %0\n s_cmp_gt_i32 %6, -1\n v_fma_f32 %1, |%9|, %3, %1\n"
KERNEL(k_v3s, REP8(EV_V3S))
// VALU + compare + branch (not taken)
#define EV_V3SB "v_add_u32 %4, %6, %7\n v_fma_f32 %0, |%9|, %2, %0\n s_cmp_gt_i32 %6, -1\n v_fma_f32 %1, |%9|, %3, %1\n s_cbranch_scc0 1f\n 1:\n"
KERNEL(k_v3sb, REP8(EV_V3SB))
// 4 places per lane: b128 read, 4 fma
#define EV_P4 "v_add_u32 %4, %6, %7\n ds_read_b64 %5, %4\n s_waitcnt lgkmcnt(3)\n v_fma_f32 %0, |%9|, %2, %0\n v_fma_f32 %1, |%9|, %3, %1\n s_cmp_gt_i32 %6, -1\n v_fma_f32 %2, |%9|, %0, %2\n v_fma_f32 %3, |%9|, %1, %3\n s_cbranch_scc0 1f\n 1:\n"
KERNEL(k_p4, REP8(EV_P4))

// control-flow costs on top of two VGPR adds per event
#define EV_NT "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n s_cmp_gt_i32 %6, -1\n s_cbranch_scc0 1f\n s_nop 0\n 1:\n"
KERNEL(k_nt, REP8(EV_NT))
#define EV_TK "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n s_cmp_gt_i32 %6, -1\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n"
KERNEL(k_tk, REP8(EV_TK))
#define EV_TK2 "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n s_cmp_gt_i32 %6, -1\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n s_branch 2f\n s_nop 0\n 2:\n"
KERNEL(k_tk2, REP8(EV_TK2))
#define EV_CALL "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n s_call_b64 s[20:21], 9f\n s_branch 8f\n 9:\n s_setpc_b64 s[20:21]\n 8:\n"
KERNEL(k_call, REP8(EV_CALL))
#define EV_A2 "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
KERNEL(k_a2, REP8(EV_A2))
// far taken branch (out of line and back), like the LAST handler
#define EV_FAR "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n s_cmp_gt_i32 %6, -1\n s_cbranch_scc1 7f\n 6:\n"
KERNEL(k_far, REP8(EV_FAR) "s_branch 5f\n 7:\n s_branch 6b\n 5:\n")

// single-spike event of the current kernel, with two scalar adds vs one packed add (w lands in a register pair)
#define EV_S2 "v_add_u32 %4, %6, %7\n ds_read_b64 %5, %4\n v_add_f32 %0, %0, %2\n s_cmp_gt_i32 %6, -1\n v_add_f32 %1, %1, %3\n s_cbranch_scc0 1f\n 1:\n"
KERNEL(k_s2, REP8(EV_S2) "s_waitcnt lgkmcnt(0)\n")
#define EV_PK "v_add_u32 %4, %6, %7\n ds_read_b64 %5, %4\n v_pk_add_f32 %[accp], %[accp], %[wp]\n s_cmp_gt_i32 %6, -1\n s_cbranch_scc0 1f\n 1:\n"
KERNEL(k_pk, REP8(EV_PK) "s_waitcnt lgkmcnt(0)\n")
// integrate-and-fire step: scalar ops vs packed ops
#define IAF_S "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n v_max_f32 %4, %0, %1\n v_cmp_le_f32 vcc, 1.0, %4\n s_cbranch_vccz 1f\n 1:\n v_fma_f32 %0, %0, 0.5, 0.5 clamp\n v_fma_f32 %1, %1, 0.5, 0.5 clamp\n v_fma_f32 %0, %0, 2.0, -1.0\n v_fma_f32 %1, %1, 2.0, -1.0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0\n"
KERNEL(k_iafs, REP8(IAF_S))
#define IAF_P "v_pk_add_f32 %[accp], %[accp], %[wp]\n v_max_f32 %4, %0, %1\n v_cmp_le_f32 vcc, 1.0, %4\n s_cbranch_vccz 1f\n 1:\n v_pk_fma_f32 %[accp], %[accp], %[half], %[half] clamp\n v_pk_fma_f32 %[accp], %[accp], %[two], %[mone]\n v_mov_b64 %[wp], 0\n"
KERNEL(k_iafp, REP8(IAF_P))

// The whole instruction mix of k_output_counts per 16-event chunk, with nothing to wait for but the LDS reads:
// 16 x (address add with an SGPR source + ds_read_b64), one lgkmcnt(0), then 16 single-spike events (two VGPR adds,
// compare, branch) of which every fourth ends a time step: taken branch to an out-of-line block (membrane adds,
// threshold test, taken branch over the spike path, 4-FMA clip, zeroing, the next event's two adds) and back.
// Its time per chunk is the issue floor of that mix; compare with the kernel's time per chunk.
#define MIX_RD "v_add_u32 %4, s20, %7\n ds_read_b64 %5, %4\n"
#define MIX_EV "v_add_f32 %0, %0, %2\n s_cmp_gt_i32 %6, -1\n v_add_f32 %1, %1, %3\n s_cbranch_scc0 90f\n"
#define MIX_LAST(n) "v_add_f32 %0, %0, %2\n s_cmp_gt_i32 %6, -1\n v_add_f32 %1, %1, %3\n s_cbranch_scc1 1" #n "f\n 2" #n ":\n"
#define MIX_STEP(n) "1" #n ":\n v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n v_max_f32 %4, %2, %3\n v_cmp_le_f32 vcc, 0x7f000000, %4\n s_cbranch_vccz 3" #n "f\n" \
                    " v_max_f32 %4, 0, %2\n v_floor_f32 %4, %4\n v_sub_f32 %2, %2, %4\n 3" #n ":\n" \
                    " v_fma_f32 %2, %2, 0.5, 0.5 clamp\n v_fma_f32 %3, %3, 0.5, 0.5 clamp\n v_fma_f32 %2, %2, 2.0, -1.0\n v_fma_f32 %3, %3, 2.0, -1.0\n" \
                    " v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n s_cmp_gt_i32 %6, -1\n s_cbranch_scc1 2" #n "b\n"
#define MIX_CHUNK "s_lshl_b32 s20, %6, 9\n" MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD \
                  "s_waitcnt lgkmcnt(0)\n" \
                  MIX_EV MIX_EV MIX_EV MIX_LAST(1) MIX_EV MIX_EV MIX_EV MIX_LAST(2) MIX_EV MIX_EV MIX_EV MIX_LAST(3) MIX_EV MIX_EV MIX_EV MIX_LAST(4) \
                  "s_branch 90f\n" MIX_STEP(1) MIX_STEP(2) MIX_STEP(3) MIX_STEP(4) "90:\n"
KERNEL(k_mix16, MIX_CHUNK)
// the same mix with the weight reads software-pipelined: wait for the reads issued one chunk earlier, issue the next
// chunk's reads, then integrate (what a planar stream layout -- offsets of chunk c+1 available while chunk c is
// integrated -- would allow)
#define MIX_RD16 MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD MIX_RD
#define MIX_EVS MIX_EV MIX_EV MIX_EV MIX_LAST(1) MIX_EV MIX_EV MIX_EV MIX_LAST(2) MIX_EV MIX_EV MIX_EV MIX_LAST(3) MIX_EV MIX_EV MIX_EV MIX_LAST(4)
#define MIX_STEPS "s_branch 90f\n" MIX_STEP(1) MIX_STEP(2) MIX_STEP(3) MIX_STEP(4) "90:\n"
KERNEL(k_mix16pre, "s_lshl_b32 s20, %6, 9\n s_waitcnt lgkmcnt(0)\n" MIX_RD16 MIX_EVS MIX_STEPS)
// ... and without any LDS read (address adds kept): the share of the reads' issue and wait
#define MIX_AD "v_add_u32 %4, s20, %7\n"
KERNEL(k_mix16nolds, "s_lshl_b32 s20, %6, 9\n" MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_AD MIX_EVS MIX_STEPS)
// ... and with the event branches never taken and no step blocks (16 plain events): the share of the time steps
#define MIX_EV16 MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV MIX_EV
KERNEL(k_mix16nostep, "s_lshl_b32 s20, %6, 9\n" MIX_RD16 "s_waitcnt lgkmcnt(0)\n" MIX_EV16 "90:\n")
// ... the step blocks inline (no taken branch into and out of them; the branch over the spike path stays)
#define MIX_STEPIN(n) "v_add_f32 %0, %0, %2\n s_cmp_gt_i32 %6, -1\n v_add_f32 %1, %1, %3\n s_cbranch_scc0 2" #n "f\n" \
                    " v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n v_max_f32 %4, %2, %3\n v_cmp_le_f32 vcc, 0x7f000000, %4\n s_cbranch_vccz 3" #n "f\n" \
                    " v_max_f32 %4, 0, %2\n v_floor_f32 %4, %4\n v_sub_f32 %2, %2, %4\n 3" #n ":\n" \
                    " v_fma_f32 %2, %2, 0.5, 0.5 clamp\n v_fma_f32 %3, %3, 0.5, 0.5 clamp\n v_fma_f32 %2, %2, 2.0, -1.0\n v_fma_f32 %3, %3, 2.0, -1.0\n" \
                    " v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n 2" #n ":\n"
KERNEL(k_mix16inl, "s_lshl_b32 s20, %6, 9\n s_waitcnt lgkmcnt(0)\n" MIX_RD16 MIX_EV MIX_EV MIX_EV MIX_STEPIN(1) MIX_EV MIX_EV MIX_EV MIX_STEPIN(2) MIX_EV MIX_EV MIX_EV MIX_STEPIN(3) MIX_EV MIX_EV MIX_EV MIX_STEPIN(4) "90:\n")

// what a hand-written loop could remove: (z) the two zeroings per step and the two adds of the event that follows a
// step (its weights ARE the new sum); (v) the address add from a VGPR-held offset (full rate) with the offsets of a
// chunk fetched by four broadcast ds_read_b128; (zv) both
#define MIX_EV0 "s_cmp_gt_i32 %6, -1\n s_cbranch_scc0 90f\n"
#define MIX_STEPZ(n) "1" #n ":\n v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n v_max_f32 %4, %2, %3\n v_cmp_le_f32 vcc, 0x7f000000, %4\n s_cbranch_vccz 3" #n "f\n" \
                    " v_max_f32 %4, 0, %2\n v_floor_f32 %4, %4\n v_sub_f32 %2, %2, %4\n 3" #n ":\n" \
                    " v_fma_f32 %2, %2, 0.5, 0.5 clamp\n v_fma_f32 %3, %3, 0.5, 0.5 clamp\n v_fma_f32 %2, %2, 2.0, -1.0\n v_fma_f32 %3, %3, 2.0, -1.0\n" \
                    " s_cmp_gt_i32 %6, -1\n s_cbranch_scc1 2" #n "b\n"
#define MIX_EVSZ MIX_EV MIX_EV MIX_EV MIX_LAST(1) MIX_EV0 MIX_EV MIX_EV MIX_LAST(2) MIX_EV0 MIX_EV MIX_EV MIX_LAST(3) MIX_EV0 MIX_EV MIX_EV MIX_LAST(4)
#define MIX_STEPSZ "s_branch 90f\n" MIX_STEPZ(1) MIX_STEPZ(2) MIX_STEPZ(3) MIX_STEPZ(4) "90:\n"
KERNEL(k_mix16z, "s_lshl_b32 s20, %6, 9\n" MIX_RD16 "s_waitcnt lgkmcnt(0)\n" MIX_EVSZ MIX_STEPSZ)
#define MIX_RDV "v_add_u32 %4, %10, %7\n ds_read_b64 %5, %4\n"
#define MIX_RDV16 MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV MIX_RDV
#define MIX_OFFS "ds_read_b128 %[accp4], %10\n ds_read_b128 %[accp4], %10 offset:16\n ds_read_b128 %[accp4], %10 offset:32\n ds_read_b128 %[accp4], %10 offset:48\n"
KERNEL4(k_mix16v, MIX_RDV16 MIX_OFFS "s_waitcnt lgkmcnt(0)\n" MIX_EVS MIX_STEPS)
KERNEL4(k_mix16zv, MIX_RDV16 MIX_OFFS "s_waitcnt lgkmcnt(0)\n" MIX_EVSZ MIX_STEPSZ)

// three places per lane (a 150 KB tile at F = 200): second address VGPR-only, ds_read_b64 + ds_read_b32, 3 adds per
// event, 3-wide step blocks.  Time per chunk covers 192 places instead of 128.
#define P3_RD "v_add_u32 %4, s20, %7\n v_add_u32 %[b2], %4, %10\n ds_read_b64 %5, %4\n ds_read_b32 %[w3], %[b2]\n"
#define P3_RD16 P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD P3_RD
#define P3_EV "v_add_f32 %0, %0, %2\n s_cmp_gt_i32 %6, -1\n v_add_f32 %1, %1, %3\n v_add_f32 %[c3], %[c3], %[w3]\n s_cbranch_scc0 90f\n"
#define P3_LAST(n) "v_add_f32 %0, %0, %2\n s_cmp_gt_i32 %6, -1\n v_add_f32 %1, %1, %3\n v_add_f32 %[c3], %[c3], %[w3]\n s_cbranch_scc1 1" #n "f\n 2" #n ":\n"
#define P3_STEP(n) "1" #n ":\n v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n v_add_f32 %[v3], %[v3], %[c3]\n v_max3_f32 %4, %2, %3, %[v3]\n v_cmp_le_f32 vcc, 0x7f000000, %4\n s_cbranch_vccnz 3" #n "f\n" \
                    " v_fma_f32 %2, %2, 0.5, 0.5 clamp\n v_fma_f32 %3, %3, 0.5, 0.5 clamp\n v_fma_f32 %[v3], %[v3], 0.5, 0.5 clamp\n v_fma_f32 %2, %2, 2.0, -1.0\n v_fma_f32 %3, %3, 2.0, -1.0\n v_fma_f32 %[v3], %[v3], 2.0, -1.0\n" \
                    " v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %[c3], 0\n s_cmp_gt_i32 %6, -1\n s_cbranch_scc1 2" #n "b\n 3" #n ":\n"
#define P3_EVS P3_EV P3_EV P3_EV P3_LAST(1) P3_EV P3_EV P3_EV P3_LAST(2) P3_EV P3_EV P3_EV P3_LAST(3) P3_EV P3_EV P3_EV P3_LAST(4)
KERNEL4(k_mix16p3, "s_lshl_b32 s20, %6, 9\n" P3_RD16 "s_waitcnt lgkmcnt(0)\n" P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")
// the two-places mix with the spike-path branch not taken (as the shipped kernel lays it out), for comparison
#define MIX_STEPN(n) "1" #n ":\n v_add_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n v_max_f32 %4, %2, %3\n v_cmp_le_f32 vcc, 0x7f000000, %4\n s_cbranch_vccnz 3" #n "f\n" \
                    " v_fma_f32 %2, %2, 0.5, 0.5 clamp\n v_fma_f32 %3, %3, 0.5, 0.5 clamp\n v_fma_f32 %2, %2, 2.0, -1.0\n v_fma_f32 %3, %3, 2.0, -1.0\n" \
                    " v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n s_cmp_gt_i32 %6, -1\n s_cbranch_scc1 2" #n "b\n 3" #n ":\n"
KERNEL4(k_mix16n, "s_lshl_b32 s20, %6, 9\n" MIX_RD16 "s_waitcnt lgkmcnt(0)\n" MIX_EVS "s_branch 90f\n" MIX_STEPN(1) MIX_STEPN(2) MIX_STEPN(3) MIX_STEPN(4) "90:\n")

// three places per lane through one 12-byte read per event (lane stride 12 bytes, 4-byte aligned)
#define P3B_RD "v_add_u32 %4, s20, %[lane12]\n ds_read_b96 %[w96], %4\n"
#define P3B_RD16 P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD P3B_RD
KERNEL4(k_mix16p3b, "s_lshl_b32 s20, %6, 9\n" P3B_RD16 "s_waitcnt lgkmcnt(0)\n" P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")

// three places per lane at lane stride 4 bytes: places l, 64+l, 128+l of a 768-byte row; one address, two reads
#define P3C_RD "v_add_u32 %4, s20, %[lane4]\n ds_read2_b32 %5, %4 offset1:64\n ds_read_b32 %[w3], %4 offset:512\n"
#define P3C_RD16 P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD P3C_RD
KERNEL4(k_mix16p3c, "s_lshl_b32 s20, %6, 9\n" P3C_RD16 "s_waitcnt lgkmcnt(0)\n" P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")

// three places per lane with the row offset taken from a VGPR instead of an SGPR
#define P3V_RD "v_add_u32 %4, %10, %7\n v_add_u32 %[b2], %4, %10\n ds_read_b64 %5, %4\n ds_read_b32 %[w3], %[b2]\n"
#define P3V_RD16 P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD P3V_RD
KERNEL4(k_mix16p3v, "s_lshl_b32 s20, %6, 9\n" P3V_RD16 "s_waitcnt lgkmcnt(0)\n" P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")

// three places per lane, the third read addressed by M0 + lane (ds_read_addtid_b32) instead of a second VGPR address
#define P3T_RD "s_lshr_b32 m0, s20, 1\n ds_read_b64 %5, %4\n ds_read_addtid_b32 %[w3]\n"
#define P3T_AD "v_add_u32 %4, s20, %7\n"
#define P3T_RD16 P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD
KERNEL4(k_mix16p3t, "s_lshl_b32 s20, %6, 9\n" P3T_RD16 "s_waitcnt lgkmcnt(0)\n" P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")
// ... and the pair addresses of two events by one 64-bit add from an SGPR pair (needs the offsets of a chunk adjacent in the stream)
#define P3U_AD "v_lshl_add_u64 %5, s[20:21], 0, %[accp]\n"
#define P3U_RD16 P3U_AD P3U_AD P3U_AD P3U_AD P3U_AD P3U_AD P3U_AD P3U_AD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD
KERNEL4(k_mix16p3u, "s_lshl_b32 s20, %6, 9\n s_mov_b32 s21, s20\n" P3U_RD16 "s_waitcnt lgkmcnt(0)\n" P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")
// ... and with no address add at all (lower bound of what the address computation costs)
#define P3N_RD16 P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD P3T_RD
KERNEL4(k_mix16p3n, "s_lshl_b32 s20, %6, 9\n" P3N_RD16 "s_waitcnt lgkmcnt(0)\n" P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")
// the shipped order for comparison: all address adds first, then the reads
#define P3S_AD "v_add_u32 %4, s20, %7\n v_add_u32 %[b2], %4, %10\n"
#define P3S_RD "ds_read_b64 %5, %4\n ds_read_b32 %[w3], %[b2]\n"
#define P3S_RD16 P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_AD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD P3S_RD
KERNEL4(k_mix16p3s, "s_lshl_b32 s20, %6, 9\n" P3S_RD16 "s_waitcnt lgkmcnt(0)\n" P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")
// two places per lane, both reads by M0 + lane
#define T2_RD "s_mov_b32 m0, s20\n s_nop 0\n ds_read_addtid_b32 %2\n ds_read_addtid_b32 %3 offset:256\n"
#define T2_RD16 T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD T2_RD
KERNEL4(k_mix16t2, "s_lshl_b32 s20, %6, 9\n" T2_RD16 "s_waitcnt lgkmcnt(0)\n" MIX_EVS "s_branch 90f\n" MIX_STEPN(1) MIX_STEPN(2) MIX_STEPN(3) MIX_STEPN(4) "90:\n")

// the shipped three-place mix (p3t) with the weight reads software-pipelined over chunks: wait for the reads issued one
// chunk earlier, issue the next chunk's reads (a second set of weight registers in the real kernel), then integrate
KERNEL4(k_mix16p3tpre, "s_lshl_b32 s20, %6, 9\n s_waitcnt lgkmcnt(0)\n" P3T_RD16 P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")
// ... and with no LDS read at all (address adds and M0 shifts kept): what the reads' issue slots cost
#define P3T_NR "s_lshr_b32 m0, s20, 1\n"
#define P3T_NR16 P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_AD P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR P3T_NR
KERNEL4(k_mix16p3tnolds, "s_lshl_b32 s20, %6, 9\n" P3T_NR16 P3_EVS "s_branch 90f\n" P3_STEP(1) P3_STEP(2) P3_STEP(3) P3_STEP(4) "90:\n")

template <typename K>
void run(const char *name, K kern) {
    unsigned long long *d; hipMalloc(&d, 512 * 16 * 8);
    struct Cfg { int blocks, threads; };
    for (Cfg c : {Cfg{256, 256}, Cfg{256, 512}, Cfg{256, 1024}, Cfg{512, 1024}}) {
        hipLaunchKernelGGL(kern, dim3(c.blocks), dim3(c.threads), 0, 0, d, 1.0f, 1u);
        { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return; } }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(c.blocks), dim3(c.threads), 0, 0, d, 1.0f, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double events = (double)ITER * 8;                     // per wave
        const double waves_per_simd = (double)c.blocks * c.threads / 64 / 1024;
        // wall time per event per SIMD, in ns (all SIMDs busy): ms*1e6 / (events * waves_per_simd)
        printf("%-8s waves/SIMD %.0f  wall ns/event/wave %.3f   ns/event/SIMD %.3f\n",
               name, waves_per_simd, ms * 1e6 / events, ms * 1e6 / (events * waves_per_simd));
    }
    hipFree(d);
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::string w = argc > 1 ? argv[1] : "all";
#define RUN(n, k) if (w == "all" || w == n) run(n, k);
    RUN("cur", k_cur) RUN("dpp", k_dpp) RUN("dpp_nolds", k_dppnl) RUN("valu3", k_v3) RUN("valu3+lds", k_v3l)
    RUN("valu3+s", k_v3s) RUN("valu3+sb", k_v3sb) RUN("ppl4", k_p4)
    RUN("ev_s2", k_s2) RUN("ev_pk", k_pk) RUN("iaf_s", k_iafs) RUN("iaf_p", k_iafp)
    RUN("mix16", k_mix16)   // "events" in its printout are whole 16-event chunks / 8
    RUN("mix16p3", k_mix16p3) RUN("mix16p3s", k_mix16p3s) RUN("mix16p3t", k_mix16p3t) RUN("mix16p3tpre", k_mix16p3tpre) RUN("mix16p3tnolds", k_mix16p3tnolds) RUN("mix16p3u", k_mix16p3u) RUN("mix16p3n", k_mix16p3n) RUN("mix16t2", k_mix16t2) RUN("mix16p3v", k_mix16p3v) RUN("mix16p3b", k_mix16p3b) RUN("mix16p3c", k_mix16p3c) RUN("mix16n", k_mix16n)
    RUN("mix16z", k_mix16z) RUN("mix16v", k_mix16v) RUN("mix16zv", k_mix16zv)
    RUN("mix16pre", k_mix16pre) RUN("mix16nolds", k_mix16nolds) RUN("mix16nostep", k_mix16nostep) RUN("mix16inl", k_mix16inl)
    RUN("add2", k_a2) RUN("add2+nt", k_nt) RUN("add2+tk", k_tk) RUN("add2+tk2", k_tk2) RUN("add2+call", k_call)
    return 0;
}

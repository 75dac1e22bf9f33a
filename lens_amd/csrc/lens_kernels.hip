// lens_kernels.hip -- gfx950 (MI355X / CDNA4) kernels and C ABI of liblens_amd.so.
//
// The LENS SNN inference-and-matching hot path, written for 64-wide wavefronts:
//   k_accumulate_events  event list -> uint8 count frames          (LDS histogram, HBM-bound)
//   k_feature_events     frames -> rate-coded raster -> feature Linear + IAF -> spike streams
//   k_output_counts      spike streams x place database (LDS weight tile) + IAF -> spike counts
//   k_seqmatch[_topk]    diagonal sequence filter (+ wave-level top-K selection)
//   k_topk_merge         merge of per-shard top-K lists
//
// Arithmetic contract: every Linear output is an ascending-input-index
// chain of f32 fused multiply-adds from +0 and the IAF update is evaluated literally in f32;
// both layers are evaluated event-driven, which leaves every chain bit-identical to the dense
// one because fmaf(0, w, acc) == acc.  Built with -ffp-contract=off: the only fused operations
// are the explicit __builtin_fmaf calls.
//
// Reference lines each kernel replaces are cited in include/lens_amd.h.
// Generated gfx950 assembly, included as asm statements: feature_loop_n2f4.inc (scripts/gen_feature_loop.py), the step
// loop of k_feature_events for the 65..128-input / 129..256-feature instantiation, and output_loop_p{1..4}.inc
// (scripts/gen_output_loop.py), the chunk loop of k_output_counts for the four LDS tile widths; fire_loop.inc
// (scripts/gen_fire_loop.py), the step walk of k_feature_fire; carry_l2_loop.inc (scripts/gen_carry_loop.py), the chunk
// loop of k_output_carry_l2.  The per-weight
// sections of the training kernels are blocks of instructions written out in this file.  Every path also exists in
// HIP C++ (other instantiations, the through-L2 variant), which documents the arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <cstdlib>
#include <atomic>
#include <mutex>

#include "../../include/lens_amd.h"

namespace {

// Spike-stream event = two 32-bit words:
//   word 0: f * ev_scale   (ev_scale = byte stride of one W2^T row in the output kernel's LDS tile, so the
//           word is the LDS byte offset of feature f's weights; 1 for the non-LDS variant).  Bit 31 of the
//           FIRST event of a chunk (EV_CHUNK events) is EV_CHUNK_MIXED: some count in the chunk is not 1.
//   word 1: float(spike count), integer valued, with
//           sign bit (EV_LAST)            = last event of its time step
//           mantissa LSB (EV_EMPTY_NEXT)  = (LAST events only, and only when count < 2^23, so that the bit is
//                                           free) the step(s) that follow carry no spikes and are not written
//   An empty step that cannot be dropped that way -- step 0, or after a count >= 2^23 -- is the explicit event
//   {0, -0.0f}.  The tail of the last chunk is padded with {0, +1.0f}: a harmless spike after the final LAST.
// Why empty steps can be dropped: after any step the membrane is a clip output, for which an input-free step
// (v += 0; fire; clip) is the identity unless v == 1.0 exactly, and then it is the identity from the second
// such step on.  v == 1.0 can only come out of the fire path, which checks EV_EMPTY_NEXT and applies that one
// spike there.  Step 0 is kept because the incoming state (carry mode) need not be a clip output.
// Why EV_CHUNK_MIXED: >99 % of the events are single spikes, for which fmaf(1, w, acc) is a plain add of two
// VGPRs -- full issue rate on gfx950, while a VALU instruction with an SGPR source (the count) issues at half.
constexpr uint32_t EV_LAST = 0x80000000u;
constexpr uint32_t EV_EMPTY_NEXT = 1u;
constexpr uint32_t EV_CHUNK_MIXED = 0x80000000u;
// Events per chunk of the output kernel = events per scalar prefetch (two s_load_dwordx16).  The scalar
// load of the next chunk is the only memory latency on the critical path (~900 cycles from L2 / Infinity
// Cache, one prefetch in flight per wavefront because LDS and scalar memory share the LGKM counter); 16
// events of compute per exposure keep four wavefronts per SIMD busy (8 events did not).
constexpr int EV_CHUNK = 16;
constexpr int EV_WORDS = 2 * EV_CHUNK;
constexpr int OUT_ORDER_MAX = 512;    // largest query group the output kernel sorts by stream length
constexpr int OUT_BOOK_BYTES = OUT_ORDER_MAX * 6 + 16;   // its lengths, order and query counter, behind the LDS tile
constexpr int OUT_BOOK_CARRY_BYTES = 16;                 // carried state: the counter word only

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// One sinabs IAF step (threshold 1, MultiSpike, MembraneSubtract, min_v_mem -1):
//   v += I ; s = (v > 0) * trunc(v) ; v -= s ; v = relu(v + 1) - 1
// floor(max(v, 0)) == (v > 0) * trunc(v) for every finite v.
__device__ __forceinline__ float iaf_update(float &v, float cur)
{
    float vm = v + cur;
    float s = __builtin_floorf(__builtin_fmaxf(vm, 0.0f));
    vm = vm - s;
    float x = __builtin_fmaxf(vm + 1.0f, 0.0f);
    v = x - 1.0f;
    return s;
}

// (iaf_update_fast below = the same update with the clip of iaf_clip.)
// The lower clip v = relu(vm + 1) - 1 in two full-rate instructions (v_max_f32 is half rate on gfx950):
//   h = clamp(fma(vm, 0.5, 0.5), 0, 1) == 0.5 * relu(fl(vm + 1))   exactly (scaling by 0.5 commutes with the
//       rounding; the upper clamp never binds because vm < 1 here, so fl(vm + 1) <= 2)
//   v = fma(h, 2, -1)                  == fl(relu(fl(vm + 1)) - 1)
__device__ __forceinline__ float iaf_clip(float vm)
{
    float h;
    asm("v_fma_f32 %0, %1, 0.5, 0.5 clamp" : "=v"(h) : "v"(vm));
    return __builtin_fmaf(h, 2.0f, -1.0f);
}

// iaf_update with the two-instruction clip (valid because vm - s < 1 always)
__device__ __forceinline__ float iaf_update_fast(float &v, float cur)
{
    float vm = v + cur;
    const float s = __builtin_floorf(__builtin_fmaxf(vm, 0.0f));
    vm = vm - s;
    v = iaf_clip(vm);
    return s;
}

// ------------------------------------------------------------------------------------------
// a1. events -> frames.  One 16-wave workgroup per frame, uint32 histogram in LDS (64 KB at 128x128:
// two workgroups = 32 waves per CU), 4 events (16 B) per lane per load.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_accumulate_events(const int16_t *__restrict__ xy,
                                                           const int64_t *__restrict__ frame_off, int roi,
                                                           uint8_t *__restrict__ frames,
                                                           unsigned long long *__restrict__ first_bad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *bins = reinterpret_cast<uint32_t *>(smem);
    const int nb = roi * roi;
    const int fr = blockIdx.x;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) bins[i] = 0u;
    __syncthreads();
    const int64_t e0 = frame_off[fr], e1 = frame_off[fr + 1];
    const uint32_t *ev = reinterpret_cast<const uint32_t *>(xy);
    unsigned long long my_bad = ~0ull;   // smallest out-of-range event seen by this thread
    auto add_event = [&](uint32_t w, int64_t e) {
        int c = (int)(int16_t)(w & 0xffffu) - 1;   // x - 1
        int r = (int)(int16_t)(w >> 16) - 1;       // y - 1
        const bool ok = (r >= -roi) & (r < roi) & (c >= -roi) & (c < roi);
        r += (r < 0) ? roi : 0;                    // Python-style negative index
        c += (c < 0) ? roi : 0;
        if (ok) atomicAdd(&bins[r * roi + c], 1u);
        else if ((unsigned long long)e < my_bad) my_bad = (unsigned long long)e;
    };
    // head up to a 16-byte boundary, body as 4 events (16 B) per lane with 4 loads in flight, tail
    int64_t ea = (e0 + 3) & ~(int64_t)3;
    if (ea > e1) ea = e1;
    const int64_t eb = ea + ((e1 - ea) & ~(int64_t)3);
    for (int64_t e = e0 + threadIdx.x; e < ea; e += blockDim.x) add_event(ev[e], e);
    const uint4 *ev4 = reinterpret_cast<const uint4 *>(ev + ea);
    const int64_t n4 = (eb - ea) >> 2;
    int64_t i = threadIdx.x;
    for (; i + 3 * (int64_t)blockDim.x < n4; i += 4 * (int64_t)blockDim.x) {
        uint4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = ev4[i + u * (int64_t)blockDim.x];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t e = ea + 4 * (i + u * (int64_t)blockDim.x);
            add_event(w[u].x, e); add_event(w[u].y, e + 1); add_event(w[u].z, e + 2); add_event(w[u].w, e + 3);
        }
    }
    for (; i < n4; i += blockDim.x) {
        const uint4 w = ev4[i];
        const int64_t e = ea + 4 * i;
        add_event(w.x, e); add_event(w.y, e + 1); add_event(w.z, e + 2); add_event(w.w, e + 3);
    }
    for (int64_t e = eb + threadIdx.x; e < e1; e += blockDim.x) add_event(ev[e], e);
    if (my_bad != ~0ull) atomicMin(first_bad, my_bad);
    __syncthreads();
    uint8_t *out = frames + (size_t)fr * nb;
    if ((nb & 3) == 0) {
        for (int i = threadIdx.x; i < (nb >> 2); i += blockDim.x) {
            const uint4 b = reinterpret_cast<const uint4 *>(bins)[i];
            reinterpret_cast<uint32_t *>(out)[i] =
                (b.x & 0xffu) | ((b.y & 0xffu) << 8) | ((b.z & 0xffu) << 16) | ((b.w & 0xffu) << 24);
        }
    } else {
        for (int i = threadIdx.x; i < nb; i += blockDim.x) out[i] = (uint8_t)(bins[i] & 0xffu);
    }
}

// ------------------------------------------------------------------------------------------
// a2-a5. frames -> feature-layer spike streams.  One wavefront per query stream.
//   NPL input registers per lane (input n = i*64 + lane), FPL feature registers per lane
//   (feature f = j*64 + lane).  USE_LDS: threshold codes and W1^T staged in LDS once per
//   workgroup and shared by its wavefronts; otherwise both are read through L1/L2.
// ------------------------------------------------------------------------------------------
struct FeatArgs {
    const uint8_t *frames; int Q; int HW;
    const int32_t *sel; int N;
    const uint8_t *codes; int T;
    const float *w1t; int F; int ldw;
    float *vfeat; int n_streams; int carry;
    uint32_t *events; int cap; uint32_t *nwords; int ev_scale;
    float *featspk;
    const uint16_t *active_steps;   // [N][256]: steps in which input n spikes for pixel value p (optional: load balance)
};

// Wavefronts per workgroup: the staged tables allow one workgroup per CU, so the workgroup is as
// large as the register budget of the instantiation allows (latency is hidden by wavefronts).
template <int NPL, int FPL> constexpr int feat_threads() { return (NPL <= 4 && FPL <= 8) ? 1024 : 256; }

template <int NPL, int FPL, bool USE_LDS>
__global__ void __launch_bounds__((feat_threads<NPL, FPL>())) k_feature_events(FeatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = NPL * 64, FP = FPL * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    uint8_t *s_codes = smem;
    float *s_w = reinterpret_cast<float *>(smem + ((a.T * NP + 15) & ~15));
    if constexpr (USE_LDS) {
        // stage the threshold codes [T][N] -> [T][NP] (padding 255 never fires) and W1^T [N][ldw] -> [N][FP]
        // (padding 0); 4 elements per thread and iteration when the row lengths allow aligned vector loads
        if ((a.N & 3) == 0 && (((uintptr_t)a.codes) & 3) == 0) {
            const int n4 = a.N >> 2;
            uint32_t *dst = reinterpret_cast<uint32_t *>(s_codes);
            const uint32_t *src = reinterpret_cast<const uint32_t *>(a.codes);
            for (int i = threadIdx.x; i < a.T * (NP / 4); i += blockDim.x) {
                const int t = i / (NP / 4), d = i - t * (NP / 4);
                dst[i] = (d < n4) ? src[(size_t)t * n4 + d] : 0xffffffffu;
            }
        } else {
            for (int i = threadIdx.x; i < a.T * NP; i += blockDim.x) {
                const int t = i / NP, n = i - t * NP;
                s_codes[i] = (n < a.N) ? a.codes[(size_t)t * a.N + n] : (uint8_t)255;
            }
        }
        if ((a.F & 3) == 0 && (a.ldw & 3) == 0 && (((uintptr_t)a.w1t) & 15) == 0) {
            const int f4n = a.F >> 2;
            float4 *dst = reinterpret_cast<float4 *>(s_w);
            for (int i = threadIdx.x; i < a.N * (FP / 4); i += blockDim.x) {
                const int n = i / (FP / 4), f4 = i - n * (FP / 4);
                dst[i] = (f4 < f4n) ? reinterpret_cast<const float4 *>(a.w1t + (size_t)n * a.ldw)[f4]
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            for (int i = threadIdx.x; i < a.N * FP; i += blockDim.x) {
                const int n = i / FP, f = i - n * FP;
                s_w[i] = (f < a.F) ? a.w1t[(size_t)n * a.ldw + f] : 0.0f;
            }
        }
    }
    int s = blockIdx.x * nw + wave;
    // Batch mode, one query per wavefront: the queries of the workgroup are dealt to its wavefronts by their work (the
    // number of input spikes, exact from the table), heaviest and lightest together, so that the four SIMDs of the CU --
    // wavefront w runs on SIMD w % 4 -- get about the same sum; the kernel's time is the slowest SIMD's.
    const bool deal = USE_LDS && a.active_steps && !a.carry && a.n_streams == a.Q && nw == 16 && !a.featspk;
    uint32_t *s_est = reinterpret_cast<uint32_t *>(smem + (((a.T * NP + 15) & ~15) + (size_t)a.N * FP * 4));   // behind the tables
    if (deal) {
        uint32_t e = 0;
        if (s < a.Q) {
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const int n = i * 64 + lane;
                if (n < a.N) e += a.active_steps[n * 256 + a.frames[(size_t)s * a.HW + a.sel[n]]];
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) e += (uint32_t)__shfl_xor((int)e, d, 64);
        if (lane == 0) s_est[wave] = e;
    }
    if constexpr (USE_LDS) __syncthreads();
    if (deal) {
        // lane j < 16 ranks wavefront j (heaviest = 0, ties by index); sorted position of my slot: SIMD g = w % 4, round
        // k = w / 4, the rounds alternate direction; I take the query of the wavefront that holds that rank
        const uint32_t mine = s_est[lane & 15];
        int r = 0;
        for (int j = 0; j < 16; ++j) {
            const uint32_t other = s_est[j];
            r += (other > mine || (other == mine && j < (lane & 15))) ? 1 : 0;
        }
        const int g = wave & 3, k = wave >> 2;
        const int want = 4 * k + ((k & 1) ? 3 - g : g);
        const uint64_t hit = __ballot(lane < 16 && r == want);
        s = blockIdx.x * nw + __builtin_ctzll(hit);
    }
    if (s >= a.n_streams) return;
    const int qs = (int)(((long long)s * a.Q) / a.n_streams);
    const int qe = (int)(((long long)(s + 1) * a.Q) / a.n_streams);

    int32_t selr[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = i * 64 + lane;
        selr[i] = (n < a.N) ? a.sel[n] : -1;
    }
    float v[FPL];
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        const int f = j * 64 + lane;
        v[j] = (a.carry && f < a.F) ? a.vfeat[(size_t)s * a.F + f] : 0.0f;
    }
    for (int q = qs; q < qe; ++q) {
        if (!a.carry) {
#pragma unroll
            for (int j = 0; j < FPL; ++j) v[j] = 0.0f;
        }
        uint32_t px[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            px[i] = (selr[i] >= 0) ? (uint32_t)a.frames[(size_t)q * a.HW + selr[i]] : 0u;
        uint32_t *ev = a.events + (size_t)q * a.cap;
        uint32_t count = 0;
        constexpr bool ASM_LOOP = (NPL == 2 && FPL == 4 && USE_LDS);
        if (ASM_LOOP && !a.featspk) {
            if constexpr (ASM_LOOP) {
                // hand-scheduled gfx950 step loop (scripts/gen_feature_loop.py); the dynamic LDS block starts at LDS
                // address 0 (no static __shared__ in this kernel): codes [T][128] bytes, then W1^T [N][256] floats
                const uint32_t w_base = ((uint32_t)a.T * NP + 15u) & ~15u;
                uint32_t caddr = (uint32_t)lane;
                uint32_t cnt = 0;
                asm volatile(
#include "feature_loop_n2f4.inc"
                    : [v0] "+&v"(v[0]), [v1] "+&v"(v[1]), [v2] "+&v"(v[2]), [v3] "+&v"(v[3]), [count] "+&s"(cnt),
                      [caddr] "+&v"(caddr)
                    : [px0] "v"(px[0]), [px1] "v"(px[1]), [wa0] "v"(w_base + (uint32_t)lane * 4u),
                      [wa1] "v"(w_base + 65536u + (uint32_t)lane * 4u),
                      [fo0] "v"((uint32_t)lane * (uint32_t)a.ev_scale), [fo1] "v"((uint32_t)(64 + lane) * (uint32_t)a.ev_scale),
                      [fo2] "v"((uint32_t)(128 + lane) * (uint32_t)a.ev_scale), [fo3] "v"((uint32_t)(192 + lane) * (uint32_t)a.ev_scale),
                      [ev] "s"(ev), [T] "s"(a.T)
                    : "memory", "vcc", "scc",
                      "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",
                      "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",
                      "s66", "s67",
                      "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",
                      "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
                count = cnt;
            }
        } else {
        uint32_t last_word1 = 0u;   // count word of the latest LAST event (wave-uniform)
        bool flagged = false;       // ... which carries EV_EMPTY_NEXT already
        auto load_codes = [&](int t, uint32_t (&code)[NPL]) {
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                if constexpr (USE_LDS) {
                    code[i] = s_codes[t * NP + i * 64 + lane];
                } else {
                    const int n = i * 64 + lane;
                    code[i] = (n < a.N) ? (uint32_t)a.codes[(size_t)t * a.N + n] : 255u;
                }
            }
        };
        uint32_t code[NPL], code_next[NPL];
        load_codes(0, code);
        for (int t = 0; t < a.T; ++t) {
            if (t + 1 < a.T) load_codes(t + 1, code_next);   // next step's thresholds: off the dependent chain
            float acc[FPL];
#pragma unroll
            for (int j = 0; j < FPL; ++j) acc[j] = 0.0f;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                uint64_t m = __ballot(px[i] >= code[i]);
                while (m) {   // active inputs in ascending n (wave-uniform loop)
                    const int n = i * 64 + __builtin_ctzll(m);
                    m &= m - 1;
#pragma unroll
                    for (int j = 0; j < FPL; ++j) {
                        float w;
                        if constexpr (USE_LDS) {
                            w = s_w[n * FP + j * 64 + lane];
                        } else {
                            const int f = j * 64 + lane;
                            w = (f < a.F) ? a.w1t[(size_t)n * a.ldw + f] : 0.0f;
                        }
                        acc[j] = acc[j] + w;   // == fmaf(1.0f, w, acc): the input spike is exactly 1
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NPL; ++i) code[i] = code_next[i];
            float sp[FPL];
            uint64_t sm[FPL];
            uint32_t total = 0;
#pragma unroll
            for (int j = 0; j < FPL; ++j) {
                sp[j] = iaf_update_fast(v[j], acc[j]);
                sm[j] = __ballot(sp[j] > 0.0f);
                total += (uint32_t)__builtin_popcountll(sm[j]);
            }
            if (a.featspk) {
#pragma unroll
                for (int j = 0; j < FPL; ++j) {
                    const int f = j * 64 + lane;
                    if (f < a.F) a.featspk[((size_t)q * a.T + t) * a.F + f] = sp[j];
                }
            }
            uint2 *ev2 = reinterpret_cast<uint2 *>(ev);
            // chunk of event e is mixed: set the flag on the chunk's first event (written in an earlier
            // instruction of this wavefront, or by the caller itself when e is that first event)
            auto mark_mixed = [&](uint32_t e) { atomicOr(&ev[2 * (e & ~(uint32_t)(EV_CHUNK - 1))], EV_CHUNK_MIXED); };
            if (total == 0) {
                if (!flagged) {
                    if (t > 0 && (last_word1 & 0x7fffffffu) < 0x4b000000u) {
                        // first empty step of a run: flag the previous LAST event (count < 2^23: its mantissa LSB is free)
                        if (lane == 0) ev[2 * (count - 1) + 1] = last_word1 | EV_EMPTY_NEXT;
                        flagged = true;
                    } else {                       // explicit empty step (see the format notes)
                        if (lane == 0) {
                            ev2[count] = make_uint2(0u, EV_LAST);
                            mark_mixed(count);
                        }
                        count += 1;
                        last_word1 = EV_LAST;
                    }
                }
            } else {
                uint32_t base = count;
                uint32_t spor = 0u;   // OR of the count bit patterns: above 1.0f's iff some count exceeds 1
#pragma unroll
                for (int j = 0; j < FPL; ++j) {
                    if (sp[j] > 0.0f) {
                        const uint32_t pos = base + lanes_below(sm[j]);
                        uint32_t bits = __float_as_uint(sp[j]);
                        if (pos == count + total - 1) bits |= EV_LAST;
                        ev2[pos] = make_uint2((uint32_t)(j * 64 + lane) * (uint32_t)a.ev_scale, bits);
                    }
                    base += (uint32_t)__builtin_popcountll(sm[j]);
                    spor |= __float_as_uint(sp[j]);
                }
                last_word1 = 0xbf800000u;               // a single spike, LAST: all but <1 % of the steps
                if (__ballot(spor > 0x3f800000u)) {     // some count exceeds 1: after the stores above, in program order
                    base = count;
#pragma unroll
                    for (int j = 0; j < FPL; ++j) {
                        if (sp[j] > 1.0f) mark_mixed(base + lanes_below(sm[j]));
                        base += (uint32_t)__builtin_popcountll(sm[j]);
                        if (sm[j])
                            last_word1 = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sp[j]), 63 - __builtin_clzll(sm[j])) | EV_LAST;
                    }
                }
                count += total;
                flagged = false;
            }
        }
        }
        // count is in events (2 words each); pad to a whole chunk of EV_CHUNK events with {0, +1.0f}
        const uint32_t pad = ((uint32_t)EV_CHUNK - (count & (EV_CHUNK - 1))) & (EV_CHUNK - 1);
        if ((uint32_t)lane < pad) reinterpret_cast<uint2 *>(ev)[count + lane] = make_uint2(0u, 0x3f800000u);
        if (lane == 0) a.nwords[q] = 2u * (count + pad);
    }
    if (a.carry) {
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const int f = j * 64 + lane;
            if (f < a.F) a.vfeat[(size_t)s * a.F + f] = v[j];
        }
    }
}

// ------------------------------------------------------------------------------------------
// a2-a5 for CARRIED state, in two passes.  The reference's query loop carries the membrane potentials from query
// to query (lens/run_model.py:234-242), so the feature layer of one stream is a chain of Q x T dependent steps --
// but only its integrate-and-fire update is: the input current of a step depends on the frame alone.
//   pass 1  k_feature_currents  (all queries in parallel, one wavefront per query): raster + centre pixels +
//           feature Linear -> I[q][t][f], the same ascending-n chain as k_feature_events;
//   pass 2  k_feature_fire      (one workgroup per carried stream, sequential in the query index): wavefront w owns
//           features [64w, 64w+64) -- they are independent of each other given I -- and walks the T steps without
//           any synchronisation: one prefetched load, the IAF update on ONE register, and on a step in which its
//           lanes fire the 64-bit spike mask into its row of an LDS table.  While the walkers go on with the next query,
//           four helper wavefronts expand the masks into the spike stream: per-step totals, a prefix sum for the
//           positions, the LAST / EMPTY_NEXT / CHUNK_MIXED flags and explicit empty steps exactly as the one-pass
//           writer of k_feature_events produces them (the streams are bit-identical); one barrier per query.
//           The currents are stored [q][t / 4][ldc][4], so that the walk fetches four steps of a feature with one 16-byte
//           load; the buffer carries FIRE_DEPTH rows of slack behind its last query: the walk prefetches that far.
// ------------------------------------------------------------------------------------------
template <int NPL, int FPL, bool USE_LDS>
__global__ void __launch_bounds__((feat_threads<NPL, FPL>())) k_feature_currents(FeatArgs a, float *__restrict__ cur, int ldc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = NPL * 64, FP = FPL * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    uint8_t *s_codes = smem;
    float *s_w = reinterpret_cast<float *>(smem + ((a.T * NP + 15) & ~15));
    if constexpr (USE_LDS) {
        for (int i = threadIdx.x; i < a.T * NP; i += blockDim.x) {
            const int t = i / NP, n = i - t * NP;
            s_codes[i] = (n < a.N) ? a.codes[(size_t)t * a.N + n] : (uint8_t)255;
        }
        for (int i = threadIdx.x; i < a.N * FP; i += blockDim.x) {
            const int n = i / FP, f = i - n * FP;
            s_w[i] = (f < a.F) ? a.w1t[(size_t)n * a.ldw + f] : 0.0f;
        }
        __syncthreads();
    }
    const int q = blockIdx.x * nw + wave;
    if (q >= a.Q) return;
    uint32_t px[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = i * 64 + lane;
        px[i] = (n < a.N) ? (uint32_t)a.frames[(size_t)q * a.HW + a.sel[n]] : 0u;
    }
    // layout [q][t / 4][ldc][4]: the four steps of a group side by side, so that the walk of pass 2 fetches a feature's
    // currents of four steps with one 16-byte load
    const int T4 = (a.T + 3) & ~3;
    float *out = cur + (size_t)q * T4 * ldc;
    for (int t0 = 0; t0 < a.T; t0 += 4) {
        float acc[4][FPL];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int t = t0 + tt;
#pragma unroll
            for (int j = 0; j < FPL; ++j) acc[tt][j] = 0.0f;
            if (t >= a.T) continue;                 // (wave-uniform) the last group of a T that is not a multiple of 4
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                uint32_t code;
                if constexpr (USE_LDS) {
                    code = s_codes[t * NP + i * 64 + lane];
                } else {
                    const int n = i * 64 + lane;
                    code = (n < a.N) ? (uint32_t)a.codes[(size_t)t * a.N + n] : 255u;
                }
                uint64_t m = __ballot(px[i] >= code);
                while (m) {   // active inputs in ascending n (wave-uniform loop)
                    const int n = i * 64 + __builtin_ctzll(m);
                    m &= m - 1;
#pragma unroll
                    for (int j = 0; j < FPL; ++j) {
                        float w;
                        if constexpr (USE_LDS) {
                            w = s_w[n * FP + j * 64 + lane];
                        } else {
                            const int f = j * 64 + lane;
                            w = (f < a.F) ? a.w1t[(size_t)n * a.ldw + f] : 0.0f;
                        }
                        acc[tt][j] = acc[tt][j] + w;   // == fmaf(1.0f, w, acc): the input spike is exactly 1
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < FPL; ++j)
            if (j * 64 < ldc)
                *reinterpret_cast<float4 *>(out + ((size_t)(t0 >> 2) * ldc + j * 64 + lane) * 4) =
                    make_float4(acc[0][j], acc[1][j], acc[2][j], acc[3][j]);
    }
}

constexpr int FIRE_DEPTH = 32;         // steps whose currents are in flight per wavefront of k_feature_fire (fire_loop.inc)
constexpr int FIRE_HELPERS = 4;        // wavefronts of k_feature_fire that expand the masks of the previous query
constexpr int FIRE_MAX_WAVES = 16 - FIRE_HELPERS;   // walking wavefronts: features <= 768
static_assert(FIRE_DEPTH == LENS_FIRE_SLACK_ROWS, "the slack the header asks for is the walk's prefetch depth");

struct FireArgs {
    const float *cur; int ldc; int Q; int T; int F;
    float *vfeat; int n_streams;
    uint32_t *events; int cap; uint32_t *nwords; int ev_scale;
    float *scratch;                     // [n_streams][2][NW][T][64] spike counts of the steps with a count above 1
};

// Inclusive prefix sum over the 64 lanes by DPP (no LDS crossbar): shifts within rows of 16 lanes, then the row totals
// broadcast to the following rows (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xa, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143 /* row_bcast:31 */, 0xc, 0xf, true);
    return x;
}

// Expansion of one query's spike masks into its stream by ONE helper wavefront (lane = time step, then lane = (step,
// walking wavefront) item); FIRE_HELPERS of them share the items and compute the positions each for itself (own copy of
// the per-step arrays: nothing to synchronise between them).
struct FireTables {
    const uint64_t *mask;     // [NW][T] lanes that fired
    const uint32_t *flag;     // [NW][T] 1: some count of the (wavefront, step) is above 1 -> counts[k][t][64]
    const float *counts;
    uint32_t *pos;            // [T + 1] this helper's own: stream position of each step
    uint32_t *tot;            // [T]     ... events of each step
    uint32_t *mixed;          // bitmap over chunks
};

__device__ __forceinline__ void fire_expand(const FireArgs &a, const FireTables &tb, int NW, int q, int h, int lane)
{
    const int T = a.T;
    const int mixed_words = (a.cap / EV_WORDS + 31) / 32;
    auto count_bits = [&](int k, int t, int l) {   // count word of lane l's event of (wavefront k, step t)
        if (!tb.flag[(size_t)k * T + t]) return 0x3f800000u;
        return __hip_atomic_load(reinterpret_cast<const uint32_t *>(tb.counts) + ((size_t)k * T + t) * 64 + l,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto step_total = [&](int t) {
        uint32_t n = 0;
        for (int k = 0; k < NW; ++k) n += (uint32_t)__builtin_popcountll(tb.mask[(size_t)k * T + t]);
        return n;
    };
    auto step_big = [&](int t) {       // count of the step's last event >= 2^23 (the step has events)
        for (int k = NW - 1; k >= 0; --k) {
            const uint64_t m = tb.mask[(size_t)k * T + t];
            if (m) return count_bits(k, t, 63 - __builtin_clzll(m)) >= 0x4b000000u;
        }
        return false;
    };
    for (int i = lane; i < mixed_words; i += 64) tb.mixed[i] = 0u;
    // sizes of the steps (their events, or 1 for an explicit empty step) and their prefix sum, 64 steps per round
    uint32_t carry = 0;
    bool any = false;                  // some step needs the chunk-marking pass
    for (int b = 0; b < T; b += 64) {
        const int t = b + lane;
        uint32_t size = 0;
        if (t < T) {
            const uint32_t tot = step_total(t);
            // an empty step that must be written as the event {0, -0.0f}: step 0, or after a count >= 2^23
            const bool expl = tot == 0 && (t == 0 || (step_total(t - 1) != 0 && step_big(t - 1)));
            tb.tot[t] = tot;
            size = tot + (expl ? 1u : 0u);
            if (tot) for (int k = 0; k < NW; ++k) any |= tb.flag[(size_t)k * T + t] != 0;
            any |= expl;
        }
        const uint32_t inc = wave_inclusive_sum(size);
        if (t < T) tb.pos[t + 1] = carry + inc;          // position after step t
        carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (lane == 0) tb.pos[0] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the arrays are read by other lanes of this wavefront below
    __builtin_amdgcn_wave_barrier();
    if (__ballot(any) != 0) {          // chunks that hold a count other than 1 (explicit empty steps, counts above 1)
        for (int t = lane; t < T; t += 64) {
            uint32_t p = tb.pos[t];
            if (tb.pos[t + 1] - p == 0) continue;
            if (tb.tot[t] == 0) { atomicOr(&tb.mixed[(p >> 4) >> 5], 1u << ((p >> 4) & 31)); continue; }
            for (int k = 0; k < NW; ++k) {
                const uint64_t m = tb.mask[(size_t)k * T + t];
                if (tb.flag[(size_t)k * T + t]) {
                    uint32_t pe = p;
                    for (uint64_t mm = m; mm; mm &= mm - 1, ++pe)
                        if (count_bits(k, t, __builtin_ctzll(mm)) > 0x3f800000u) atomicOr(&tb.mixed[(pe >> 4) >> 5], 1u << ((pe >> 4) & 31));
                }
                p += (uint32_t)__builtin_popcountll(m);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t *ev = a.events + (size_t)q * a.cap;
    // one item per (step, walking wavefront): that wavefront's events of the step, behind those of the lower ones;
    // the helpers take the items in turns of 64
    for (int i = h * 64 + lane; i < T * NW; i += FIRE_HELPERS * 64) {
        const int t = i / NW, k = i - t * NW;
        uint32_t p = tb.pos[t];
        if (tb.pos[t + 1] - p == 0) continue;
        // the step(s) that follow carry no spikes and are not written: flag on this step's LAST event
        const bool empty_next = (t + 1 < T) && (tb.pos[t + 2] - tb.pos[t + 1] == 0);
        auto put = [&](uint32_t pos, uint32_t w0, uint32_t w1) {
            if ((pos & (EV_CHUNK - 1)) == 0 && ((tb.mixed[(pos >> 4) >> 5] >> ((pos >> 4) & 31)) & 1u)) w0 |= EV_CHUNK_MIXED;
            reinterpret_cast<uint2 *>(ev)[pos] = make_uint2(w0, w1);
        };
        const uint32_t tot = tb.tot[t];
        if (tot == 0) {                // explicit empty step
            if (k == 0) put(p, 0u, EV_LAST | (empty_next ? EV_EMPTY_NEXT : 0u));
            continue;
        }
        uint64_t m = tb.mask[(size_t)k * T + t];
        if (!m) continue;
        const uint32_t last = p + tot - 1;
        for (int j = 0; j < k; ++j) p += (uint32_t)__builtin_popcountll(tb.mask[(size_t)j * T + t]);
        const bool counted = tb.flag[(size_t)k * T + t] != 0;      // counts above 1 in this (wavefront, step): rare
        for (; m; m &= m - 1) {
            const int l = __builtin_ctzll(m);
            uint32_t w1 = counted ? count_bits(k, t, l) : 0x3f800000u;
            if (p == last) w1 |= EV_LAST | ((empty_next && (w1 & 0x7fffffffu) < 0x4b000000u) ? EV_EMPTY_NEXT : 0u);
            put(p, (uint32_t)(k * 64 + l) * (uint32_t)a.ev_scale, w1);
            ++p;
        }
    }
    if (h == 0) {
        const uint32_t count = tb.pos[T];
        const uint32_t pad = ((uint32_t)EV_CHUNK - (count & (EV_CHUNK - 1))) & (EV_CHUNK - 1);
        if ((uint32_t)lane < pad) reinterpret_cast<uint2 *>(ev)[count + lane] = make_uint2(0u, 0x3f800000u);
        if (lane == 0) a.nwords[q] = 2u * (count + pad);
    }
}

__global__ void __launch_bounds__(1024) k_feature_fire(FireArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Workgroup = NW walking wavefronts (one per 64 features) + FIRE_HELPERS helper wavefronts.  Software pipeline over
    // the stream's queries, ONE barrier per iteration: in iteration i the walkers walk query i while the helpers expand
    // the masks of query i - 1 into its stream, so the expansion costs the walk nothing as long as it is the shorter.
    const int NW = (a.F + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int T = a.T;
    // per (wavefront, step): the 64-bit mask of the lanes that fired (0: no event) and a flag word (1: some count of the
    // step is above 1; the 64 counts are then in the wavefront's scratch row of the step) -- two sets, used by
    // alternate queries; per helper: positions and sizes of the steps, bitmap of the chunks with a count other than 1
    const size_t set_bytes = ((size_t)NW * T * 12 + 7) & ~(size_t)7;
    const int mixed_words = (a.cap / EV_WORDS + 31) / 32;
    const size_t helper_words = (size_t)(2 * T + 1) + mixed_words;
    const int s = blockIdx.x;
    const int qs = (int)(((long long)s * a.Q) / a.n_streams);
    const int qe = (int)(((long long)(s + 1) * a.Q) / a.n_streams);
    const int f = w * 64 + lane;
    const bool walker = w < NW;
    float v = (walker && f < a.F) ? a.vfeat[(size_t)s * a.F + f] : 0.0f;
    const float *counts_sets = a.scratch + (size_t)s * 2 * NW * ((size_t)T * 64);   // [2][NW][T][64] spike counts of flagged steps
    using lds_ptr = __attribute__((address_space(3))) unsigned char *;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
    for (int it = 0; it <= qe - qs; ++it) {
        if (walker && it < qe - qs) {
            // ---------------- the T steps of query qs + it, every walker on its own (it touches only its rows of the tables)
            const int q = qs + it;
            const size_t set = (size_t)(it & 1) * set_bytes;
            uint64_t *s_mask = reinterpret_cast<uint64_t *>(smem + set);                        // [NW][T]
            uint32_t *s_flag = reinterpret_cast<uint32_t *>(smem + set + (size_t)NW * T * 8);   // [NW][T]
            for (int t = lane; t < T; t += 64) { s_mask[(size_t)w * T + t] = 0ull; s_flag[(size_t)w * T + t] = 0u; }
            const float *row0 = a.cur + (size_t)q * ((T + 3) & ~3) * a.ldc;   // wave-uniform; [q][t / 4][ldc][4]
            const float *sc = counts_sets + ((size_t)(it & 1) * NW + w) * ((size_t)T * 64);
            uint64_t hot = ~0ull;         // lanes whose membrane is exactly 1.0; all ones: step 0 always runs
            uint32_t tab = lds0 + (uint32_t)(set + ((size_t)w * T) * 8);
            uint32_t flg = lds0 + (uint32_t)(set + (size_t)NW * T * 8 + ((size_t)w * T) * 4);
            // hand-scheduled gfx950 walk (scripts/gen_fire_loop.py): per step  v += I; fire test at 1 - 2^-24;
            // s = floor(max(v, 0)); v -= s; v = clip(v)  -- the f32 operations of iaf_update_fast in its order
            asm volatile(
#include "fire_loop.inc"
                : [v] "+&v"(v), [hot] "+&s"(hot), [tab] "+&v"(tab), [flg] "+&v"(flg)
                : [ptr] "s"(row0), [sc] "s"(sc), [off] "v"((uint32_t)f * 16u), [lane4] "v"((uint32_t)lane * 4u),
                  [ldc4] "s"((uint32_t)a.ldc * 16u), [T] "s"(T)
                : "memory", "vcc", "scc",
                  "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
                  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
                  "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
                  "v96", "v97", "v98", "v99");
        } else if (!walker && it >= 1) {
            // ---------------- the stream of query qs + it - 1 from the other set of tables
            const int h = w - NW;
            const size_t set = (size_t)((it - 1) & 1) * set_bytes;
            uint32_t *mine = reinterpret_cast<uint32_t *>(smem + 2 * set_bytes) + (size_t)h * helper_words;
            FireTables tb{reinterpret_cast<const uint64_t *>(smem + set),
                          reinterpret_cast<const uint32_t *>(smem + set + (size_t)NW * T * 8),
                          counts_sets + (size_t)((it - 1) & 1) * NW * ((size_t)T * 64),
                          mine, mine + (T + 1), mine + (2 * T + 1)};
            fire_expand(a, tb, NW, qs + it - 1, h, lane);
        }
        // The tables (LDS) and the rare count rows (global: the walk ends with s_waitcnt vmcnt(0), the helpers read them
        // past their L1 with agent-scope atomic loads) of query i are complete for the helpers of iteration i + 1, and the
        // set that the helpers have just read is free for the walk of query i + 1.
        __syncthreads();
    }
    if (walker && f < a.F) a.vfeat[(size_t)s * a.F + f] = v;
}

// ------------------------------------------------------------------------------------------
// a5-a6. spike streams -> output spike counts.
//   A workgroup owns a tile of PT = 64*PPL places (their W2^T columns staged in LDS once) and a
//   group of queries; each wavefront integrates whole queries against the tile, one lane =
//   PPL places (lane_place), membrane potentials and spike counters in registers, the T-step
//   loop inside the kernel.  The stream is read with scalar loads, 16 words at a time.
// ------------------------------------------------------------------------------------------
struct OutArgs {
    const uint32_t *events; int cap; const uint32_t *nwords; int Q;
    const float *w2t; int F; int P; int ldw;
    float *vout; int ldv; int n_groups; int carry;     // carry: 1 = one wavefront per (tile, stream), 2 = 16 streams per tile
    float *counts; int ldo;
    int ntiles;
    int n_whole;     // batch mode: workgroups of the main part (whole rounds of the CUs), 0 = none
    int n_main;      // ... tiles per round of the main part
    int g_main;      // ... query groups per tile there (1: a workgroup takes a whole tile)
    int row_mul;     // tiles read through L2: bytes of w2t per unit of an event's offset word (4 * ldw / event scale)
};

template <int PPL> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

// Place (within the tile) of component k of a lane.  PPL = 3 splits the 192-place tile row into a 128-place part read
// as one 8-byte word per lane and a 64-place part read as one 4-byte word per lane (a 12-byte read per lane would not
// be aligned and runs at a fraction of the LDS rate).  In LDS the two parts are separate planes: [F][64] floats of
// the third places at the start of the workgroup's LDS, then [F][128] floats of the pairs; an event word (row x 512)
// addresses the pairs directly and the thirds after a shift by one (ds_read_addtid_b32: M0 + 4 x lane, no vector
// address).
template <int PPL>
__device__ __forceinline__ int lane_place(int lane, int k)
{
    if constexpr (PPL == 3) return k < 2 ? 2 * lane + k : 128 + lane;
    else return lane * PPL + k;
}

template <int PPL>
__device__ __forceinline__ void vec_unpack(const typename VecT<PPL>::type &x, float (&o)[PPL])
{
    if constexpr (PPL == 1) { o[0] = x; }
    else if constexpr (PPL == 2) { o[0] = x.x; o[1] = x.y; }
    else { o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w; }
}

// Weight reads for one chunk of the spike stream, issued together.  They are completed (weights_token)
// before the NEXT chunk's scalar load is put in flight: LDS and scalar memory share the LGKM counter and
// scalar loads return out of order, so an LDS wait with a scalar load outstanding would degenerate to
// lgkmcnt(0) and wait for the prefetch too.
template <int PPL, bool W_LDS>
__device__ __forceinline__ void load_weights(const uint32_t (&e)[EV_WORDS], const unsigned char *smem, const float *w2t,
                                             int row_mul, int pl, uint32_t lane_off, float (&w)[EV_CHUNK][PPL])
{
    static_assert(PPL != 3, "three places per lane exist in the assembly loop only");
#pragma unroll
    for (int i = 0; i < EV_CHUNK; ++i) {
        const uint32_t off = i ? e[2 * i] : (e[0] & ~EV_CHUNK_MIXED);
        using vec_t = typename VecT<PPL>::type;
        vec_t wv;
        if constexpr (W_LDS) {
            wv = *reinterpret_cast<const vec_t *>(smem + (off + lane_off));
        } else {
            wv = *reinterpret_cast<const vec_t *>(reinterpret_cast<const char *>(w2t + pl) + (size_t)off * (uint32_t)row_mul);
        }
        vec_unpack<PPL>(wv, w[i]);
    }
}

// Ordering token without any memory clobber or `volatile` (either would stop the compiler from using
// scalar loads for the stream): a zero in an SGPR that is *data-dependent* on the weights just read.
// Adding it to the next chunk's offset makes the next scalar load wait until the LDS reads have
// completed (the asm needs the weights in registers => lgkmcnt(0) before it).
template <int PPL>
__device__ __forceinline__ uint32_t weights_token(const float (&w)[EV_CHUNK][PPL])
{
    // one read instruction per event returns all components, so the first component of every event covers them
    uint32_t tok;
    static_assert(EV_CHUNK == 16, "operand list below");
    asm("s_mov_b32 %0, 0" : "=s"(tok)
        : "v"(w[0][0]), "v"(w[1][0]), "v"(w[2][0]), "v"(w[3][0]), "v"(w[4][0]), "v"(w[5][0]), "v"(w[6][0]), "v"(w[7][0]),
          "v"(w[8][0]), "v"(w[9][0]), "v"(w[10][0]), "v"(w[11][0]), "v"(w[12][0]), "v"(w[13][0]), "v"(w[14][0]), "v"(w[15][0]));
    return tok;
}

// The events of a chunk against this lane's PPL places.  All control flow is wave-uniform (the
// event words live in SGPRs); no memory instruction in here.
template <int PPL, bool SINGLE>   // SINGLE: every count of the chunk is 1 (EV_CHUNK_MIXED clear)
__device__ __forceinline__ void integrate_chunk(const uint32_t (&e)[EV_WORDS], const float (&w)[EV_CHUNK][PPL],
                                                float (&v)[PPL], float (&cnt)[PPL], float (&acc)[PPL])
{
#pragma unroll
    for (int i = 0; i < EV_CHUNK; ++i) {
        // The flag bit EV_EMPTY_NEXT exists only on counts below 2^23 (whose mantissa LSB is zero); from 2^23 on the
        // LSB is part of the count (never the case on a SINGLE chunk: all its counts are 1).
        if constexpr (SINGLE) {
#pragma unroll
            for (int k = 0; k < PPL; ++k) acc[k] = acc[k] + w[i][k];   // == fmaf(1, w, acc)
        } else {
            uint32_t cbits = e[2 * i + 1] & 0x7ffffffeu;               // sign (LAST) and flag bit cleared
            if (__builtin_expect(cbits >= 0x4b000000u, 0)) cbits = e[2 * i + 1] & 0x7fffffffu;
            const float sv = __uint_as_float(cbits);
#pragma unroll
            for (int k = 0; k < PPL; ++k) acc[k] = __builtin_fmaf(sv, w[i][k], acc[k]);
        }
        if ((int32_t)e[2 * i + 1] < 0) {   // LAST event of the step: integrate-and-fire
            bool fire = false;
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                v[k] = v[k] + acc[k];
                acc[k] = 0.0f;
                fire |= (v[k] >= 0.99999994f);   // 1 - 2^-24 included: the clip rounds it up to exactly 1.0 (EV_EMPTY_NEXT)
            }
            // some lane reaches threshold: ~13 % of the steps of a 128-place tile at the benchmark's spike density, so the
            // spike path is laid out of line (a taken branch costs ~2 ns of SIMD time here, a not-taken one ~0.3)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(fire) != 0, 0)) {
#pragma unroll
                for (int k = 0; k < PPL; ++k) {
                    const float sp = __builtin_floorf(__builtin_fmaxf(v[k], 0.0f));
                    v[k] -= sp;
                    cnt[k] += sp;
                }
                // dropped empty steps follow: a membrane left at exactly 1.0 fires there
                if ((e[2 * i + 1] & EV_EMPTY_NEXT) && (SINGLE || (e[2 * i + 1] & 0x7ffffffeu) < 0x4b000000u)) {
#pragma unroll
                    for (int k = 0; k < PPL; ++k) {
                        if (iaf_clip(v[k]) == 1.0f) { cnt[k] += 1.0f; v[k] = 0.0f; }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < PPL; ++k) v[k] = iaf_clip(v[k]);
        }
    }
}

// The chunk loop of one query in gfx950 assembly (scripts/gen_output_loop.py), for the LDS tiles: same f32 operations in
// the same order as integrate_chunk, laid out as two chains of code that C++ cannot express without flag bookkeeping.
// `nwords` = address of the stream's length in words (loaded together with the first chunk: whole chunks are
// readable from any stream, the workspace holds the worst case), `lane_addr` = LDS byte address of this lane's first
// weights in row 0 of the tile (three places per lane: of its pair of places; the third place is read through M0 from
// the plane at LDS address 0, see lane_place).
#define LENS_OUT_SGPRS                                                                                                   \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",      \
    "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",      \
    "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",      \
    "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99" LENS_OUT_JUMP_SGPRS
#define LENS_OUT_V64_79 "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"
#define LENS_OUT_V80_95 "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95"
#define LENS_OUT_V96_111                                                                                                 \
    "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111"
#define LENS_OUT_V112_127                                                                                                \
    "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
// -DLENS_OUTPUT_JUMP: the operands of the computed-jump variant of the loop (gen_output_loop.py with LENS_GEN_JUMP=1;
// measured slower than the loop with a test per event, kept as a reproducible experiment, not built by default)
#ifdef LENS_OUTPUT_JUMP
#define LENS_OUT_JUMP_OUT                                                                                                \
    , [m] "=&s"(lastm), [x] "=&s"(jx), [bl] "=&s"(pc_lo), [bh] "=&s"(pc_hi), [vw] "=&v"(vwords), [va] "=&v"(vaddr)
#define LENS_OUT_JUMP_IN , [voff] "v"(voff)
#define LENS_OUT_JUMP_SGPRS , "s100", "s101"
#else
#define LENS_OUT_JUMP_OUT
#define LENS_OUT_JUMP_IN
#define LENS_OUT_JUMP_SGPRS
#endif
#define LENS_OUT_COMMON_OUT                                                                                              \
    [soff] "+&s"(soff), [nch] "=&s"(nch), [t] "=&v"(t), [t2] "=&v"(t2), [sc] "=&s"(sc), [so] "=&s"(so) LENS_OUT_JUMP_OUT
#define LENS_OUT_COMMON_IN [ev] "s"(ev), [nwp] "s"(nwords), [lane] "v"(lane_addr), [thr] "v"(thr) LENS_OUT_JUMP_IN
template <int PPL>
__device__ __forceinline__ void output_loop_asm(const uint32_t *ev, const uint32_t *nwords, uint32_t lane_addr,
                                                float (&v)[PPL], float (&cnt)[PPL])
{
    static_assert(PPL >= 1 && PPL <= 4, "generated for one to four places per lane");
    uint32_t soff = 0, nch, sc, so;
    float a[PPL], t, t2;
    const float thr = 0.99999994f;   // 1 - 2^-24: the clip rounds it up to exactly 1.0
#ifdef LENS_OUTPUT_JUMP
    uint32_t lastm, jx, pc_lo, pc_hi, vwords, vaddr;
    const uint32_t voff = (threadIdx.x & 15u) * 8u + 4u;   // byte offset of count word (lane & 15) in a chunk
#endif
    if constexpr (PPL == 1) {
        asm volatile(
#include "output_loop_p1.inc"
            : [v0] "+&v"(v[0]), [c0] "+&v"(cnt[0]), [a0] "=&v"(a[0]), LENS_OUT_COMMON_OUT
            : LENS_OUT_COMMON_IN
            : "memory", "vcc", "scc", LENS_OUT_SGPRS, LENS_OUT_V64_79);
    } else if constexpr (PPL == 2) {
        asm volatile(
#include "output_loop_p2.inc"
            : [v0] "+&v"(v[0]), [v1] "+&v"(v[1]), [c0] "+&v"(cnt[0]), [c1] "+&v"(cnt[1]), [a0] "=&v"(a[0]), [a1] "=&v"(a[1]),
              LENS_OUT_COMMON_OUT
            : LENS_OUT_COMMON_IN
            : "memory", "vcc", "scc", LENS_OUT_SGPRS, LENS_OUT_V64_79, LENS_OUT_V80_95);
    } else if constexpr (PPL == 3) {
        asm volatile(
#include "output_loop_p3.inc"
            : [v0] "+&v"(v[0]), [v1] "+&v"(v[1]), [v2] "+&v"(v[2]), [c0] "+&v"(cnt[0]), [c1] "+&v"(cnt[1]), [c2] "+&v"(cnt[2]),
              [a0] "=&v"(a[0]), [a1] "=&v"(a[1]), [a2] "=&v"(a[2]), LENS_OUT_COMMON_OUT
            : LENS_OUT_COMMON_IN
            : "memory", "vcc", "scc", "m0", LENS_OUT_SGPRS, LENS_OUT_V64_79, LENS_OUT_V80_95, LENS_OUT_V96_111);
    } else {
        asm volatile(
#include "output_loop_p4.inc"
            : [v0] "+&v"(v[0]), [v1] "+&v"(v[1]), [v2] "+&v"(v[2]), [v3] "+&v"(v[3]), [c0] "+&v"(cnt[0]), [c1] "+&v"(cnt[1]),
              [c2] "+&v"(cnt[2]), [c3] "+&v"(cnt[3]), [a0] "=&v"(a[0]), [a1] "=&v"(a[1]), [a2] "=&v"(a[2]), [a3] "=&v"(a[3]),
              LENS_OUT_COMMON_OUT
            : LENS_OUT_COMMON_IN
            : "memory", "vcc", "scc", LENS_OUT_SGPRS, LENS_OUT_V64_79, LENS_OUT_V80_95, LENS_OUT_V96_111, LENS_OUT_V112_127);
    }
}
#undef LENS_OUT_SGPRS
#undef LENS_OUT_V64_79
#undef LENS_OUT_V80_95
#undef LENS_OUT_V96_111
#undef LENS_OUT_V112_127
#undef LENS_OUT_COMMON_OUT
#undef LENS_OUT_JUMP_OUT
#undef LENS_OUT_JUMP_IN
#undef LENS_OUT_JUMP_SGPRS
#undef LENS_OUT_COMMON_IN

// LONE: the instantiation for carried state with one wavefront per (tile, stream) (a.carry == 1 on an LDS tile): kept
// apart so that its staging code does not touch the batch kernel's code generation.
template <int PPL, bool W_LDS, bool LONE = false>
__global__ void __launch_bounds__(1024) k_output_counts(OutArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PT = 64 * PPL;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    // Carried state: workgroup = (tile, stream), ONE wavefront, which walks the stream's queries in order.
    // Batch mode: the first n_whole workgroups form whole rounds of the CUs, one workgroup per CU: each takes one tile
    // and all Q queries (or, for a database of fewer tiles than CUs, the 1/2, 1/4 ... of them that makes the round
    // full): the tile is staged once, the 16 wavefronts pull queries first come, first served (stream lengths vary
    // by +-20 % (std) between queries; with a fixed assignment the slowest wavefront sets the pace) and idle only at
    // the very end, and all workgroups of a round walk their queries at the same pace, so a stream is fetched from
    // memory once per XCD and round and found in that L2 by the other workgroups.  The remaining tiles (fewer than a
    // round) are split into n_groups query groups each so that they fill the CUs once more: workgroup = (tile, group),
    // group-major.
    // The kernel owns no static LDS: the tile starts at LDS address 0 (the assembly loop of the 192-place tile relies on
    // it, launch_output checks it) and the bookkeeping sits behind the tile in the same dynamic block.
    unsigned char *book = smem + (W_LDS ? (size_t)a.F * PT * 4 : 0);
    uint32_t *s_len = reinterpret_cast<uint32_t *>(book);                              // [OUT_ORDER_MAX]
    uint16_t *s_order = reinterpret_cast<uint16_t *>(book + OUT_ORDER_MAX * 4);          // longest stream first (when the group is small enough to rank)
    int &s_next = *reinterpret_cast<int *>(book + (a.carry ? 0 : OUT_ORDER_MAX * 6));   // carried state: the book is this word only

    int tile, grp = 0, qs = 0, qe = a.Q;
    bool active = true;
    if (a.carry == 2) {
        // carried state, shared tile: workgroup = (tile, 16 streams), wavefront w walks stream 16 * group + w in order
        tile = blockIdx.x % a.ntiles;
        grp = (blockIdx.x / a.ntiles) * 16 + wave;
        active = grp < a.n_groups;
        qs = active ? (int)(((long long)grp * a.Q) / a.n_groups) : 0;
        qe = active ? (int)(((long long)(grp + 1) * a.Q) / a.n_groups) : 0;
    } else if ((int)blockIdx.x < a.n_whole) {
        // a round of the main part = n_main tiles x g_main query groups = one workgroup per CU, group-major (neighbouring
        // workgroups, which land on different XCDs, share a group's queries); g_main = 1 where the tiles alone fill the CUs
        const int per_round = a.n_main * a.g_main, j = blockIdx.x % per_round;
        tile = (blockIdx.x / per_round) * a.n_main + j % a.n_main;
        grp = j / a.n_main;
        qs = (int)(((long long)grp * a.Q) / a.g_main);
        qe = (int)(((long long)(grp + 1) * a.Q) / a.g_main);
    } else {
        const int first = (a.n_whole / (a.n_main * a.g_main)) * a.n_main;     // tiles of the main part
        const int r = blockIdx.x - a.n_whole, rem = a.ntiles - first;
        tile = first + r % rem;
        grp = r / rem;
        qs = (int)(((long long)grp * a.Q) / a.n_groups);
        qe = (int)(((long long)(grp + 1) * a.Q) / a.n_groups);
    }
    const uint32_t lane_off = (uint32_t)lane * (PPL == 3 ? 8 : PPL * 4);
    {
        const int nq = qe - qs;
        const int p0 = tile * PT;
        if (threadIdx.x == 0) s_next = 0;
        if constexpr (W_LDS) {
            float4 *dst = reinterpret_cast<float4 *>(smem);
            const int n4 = a.F * (PT / 4);
            // carried state, one wavefront per tile: eight independent 16-byte loads in flight per thread, then their stores
            // (one load and one store at a time cost such a workgroup ~10 us per launch, and a pipelined stream launches
            // every 16 queries); the batch mode keeps the plain loop below -- its staging is amortised over thousands of queries
            constexpr int SU = 8;
            for (int i0 = threadIdx.x; LONE && i0 < n4; i0 += blockDim.x * SU) {
                float4 x[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i = i0 + u * blockDim.x;
                    const int f = i / (PT / 4), c = i - f * (PT / 4);
                    x[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    // (a 192-place tile may reach past the padded row: ldw is a multiple of 256, not of 192)
                    if (i < n4 && (PPL != 3 || p0 + c * 4 < a.ldw))
                        x[u] = *reinterpret_cast<const float4 *>(a.w2t + (size_t)f * a.ldw + p0 + c * 4);
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i = i0 + u * blockDim.x;
                    if (i >= n4) continue;
                    if constexpr (PPL == 3) {
                        // places 0..127 of the row -> pairs plane (behind the F x 256 bytes of the thirds plane), 128..191 -> thirds
                        const int f = i / (PT / 4), c = i - f * (PT / 4);
                        unsigned char *d = c < 32 ? smem + (size_t)a.F * 256 + (size_t)f * 512 + c * 16 : smem + (size_t)f * 256 + (c - 32) * 16;
                        *reinterpret_cast<float4 *>(d) = x[u];
                    } else {
                        dst[i] = x[u];
                    }
                }
            }
            for (int i = threadIdx.x; !LONE && i < n4; i += blockDim.x) {
                const int f = i / (PT / 4), c = i - f * (PT / 4);
                if constexpr (PPL == 3) {
                    float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (p0 + c * 4 < a.ldw) x = *reinterpret_cast<const float4 *>(a.w2t + (size_t)f * a.ldw + p0 + c * 4);
                    unsigned char *d = c < 32 ? smem + (size_t)a.F * 256 + (size_t)f * 512 + c * 16 : smem + (size_t)f * 256 + (c - 32) * 16;
                    *reinterpret_cast<float4 *>(d) = x;
                } else {
                    dst[i] = *reinterpret_cast<const float4 *>(a.w2t + (size_t)f * a.ldw + p0 + c * 4);
                }
            }
        }
        // Longest first pays when the whole grid is one or two rounds of workgroups (the tail of the slowest one is the
        // kernel's tail); with many rounds the dispatcher evens things out and the O(nq^2) ranking is a net loss.
        const bool ranked = !a.carry && nq <= OUT_ORDER_MAX && gridDim.x <= 512;
        if (ranked) {
            for (int t = threadIdx.x; t < nq; t += blockDim.x) s_len[t] = a.nwords[qs + t];
        }
        __syncthreads();
        if (ranked) {   // rank by counting: position of query t = number of queries that come before it
            for (int t = threadIdx.x; t < nq; t += blockDim.x) {
                const uint32_t mine = s_len[t];
                int r = 0;
                for (int j = 0; j < nq; ++j) {
                    const uint32_t other = s_len[j];
                    r += (other > mine || (other == mine && j < t)) ? 1 : 0;
                }
                s_order[r] = (uint16_t)t;
            }
            __syncthreads();
        }
        // a carried stream is sequential: one wavefront per (tile, stream) walks it; the workgroup's other wavefronts only
        // helped to stage the tile (a lone wavefront takes ~15 us for 50 KB, every launch of a pipelined stream again)
        if (a.carry == 1 && wave != 0) return;
        const int pl = p0 + lane * PPL;   // first place of this lane (contiguous layouts)
        int place[PPL];
#pragma unroll
        for (int k = 0; k < PPL; ++k) place[k] = p0 + lane_place<PPL>(lane, k);

        float v[PPL], cnt[PPL], acc[PPL];
#pragma unroll
        for (int k = 0; k < PPL; ++k) v[k] = 0.0f;
        if (a.carry) {
#pragma unroll
            for (int k = 0; k < PPL; ++k) v[k] = (active && place[k] < a.P) ? a.vout[(size_t)grp * a.ldv + place[k]] : 0.0f;
        }
        for (int it = 0;; ++it) {
            int q = qs + it;                       // carry: the one wavefront walks the run in order
            if (!a.carry) {
                int t = 0;
                if (lane == 0) t = atomicAdd(&s_next, 1);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t >= nq) break;
                q = qs + (ranked ? (int)s_order[t] : t);
            }
            if (q >= qe) break;
            q = __builtin_amdgcn_readfirstlane(q);   // wave-uniform by construction; the stream pointer must live in SGPRs
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                if (!a.carry) v[k] = 0.0f;
                cnt[k] = 0.0f;
                acc[k] = 0.0f;
            }
            const uint32_t *ev = a.events + (size_t)q * a.cap;
            if constexpr (W_LDS) {
                using lds_ptr = __attribute__((address_space(3))) unsigned char *;
                const uint32_t lane_addr = (uint32_t)(uintptr_t)(lds_ptr)smem + lane_off + (PPL == 3 ? (uint32_t)a.F * 256u : 0u);
                output_loop_asm<PPL>(ev, a.nwords + q, lane_addr, v, cnt);
            } else {
                const int nch = __builtin_amdgcn_readfirstlane((int)(a.nwords[q] / EV_WORDS));
                // One chunk (EV_CHUNK events) per scalar prefetch, two SGPR buffers in ping-pong.  Per chunk: weight
                // reads -> lgkmcnt(0) -> scalar load of the next chunk -> VALU/SALU-only integration while it is in flight.
                uint32_t ea[EV_WORDS], eb[EV_WORDS];
                float w[EV_CHUNK][PPL];
#pragma unroll
                for (int i = 0; i < EV_WORDS; ++i) ea[i] = ev[i];
                const char *p = reinterpret_cast<const char *>(ev);   // chunk being integrated (byte pointer: the token adds unshifted)
                for (int c = 0; c < nch; c += 2) {
                    load_weights<PPL, W_LDS>(ea, smem, a.w2t, a.row_mul, pl, lane_off, w);
                    p += EV_WORDS * 4;
                    const uint32_t *nxt = reinterpret_cast<const uint32_t *>(p + weights_token<PPL>(w));
                    if (c + 1 < nch) {
#pragma unroll
                        for (int i = 0; i < EV_WORDS; ++i) eb[i] = nxt[i];
                    }
                    if (ea[0] & EV_CHUNK_MIXED) integrate_chunk<PPL, false>(ea, w, v, cnt, acc);
                    else integrate_chunk<PPL, true>(ea, w, v, cnt, acc);
                    if (c + 1 >= nch) break;
                    load_weights<PPL, W_LDS>(eb, smem, a.w2t, a.row_mul, pl, lane_off, w);
                    p += EV_WORDS * 4;
                    nxt = reinterpret_cast<const uint32_t *>(p + weights_token<PPL>(w));
                    if (c + 2 < nch) {
#pragma unroll
                        for (int i = 0; i < EV_WORDS; ++i) ea[i] = nxt[i];
                    }
                    if (eb[0] & EV_CHUNK_MIXED) integrate_chunk<PPL, false>(eb, w, v, cnt, acc);
                    else integrate_chunk<PPL, true>(eb, w, v, cnt, acc);
                }
            }
            float *o = a.counts + (size_t)q * a.ldo;
#pragma unroll
            for (int k = 0; k < PPL; ++k)
                if (place[k] < a.P) o[place[k]] = cnt[k];
        }
        if (a.carry && active) {
#pragma unroll
            for (int k = 0; k < PPL; ++k)
                if (place[k] < a.P) a.vout[(size_t)grp * a.ldv + place[k]] = v[k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// a5-a6, carried state with the database read through L2: one wavefront per (64-place tile, stream), no LDS tile, so
// every tile of a large database is resident at once (three 50 KB LDS tiles per CU make 100 000 places three rounds of
// 768 wavefronts, and the rounds of a sequential stream add up).  The lone wavefront runs at the latency of its own
// chain; its chunk loop (scripts/gen_carry_loop.py) keeps the weight rows of four chunks in flight.  Tiling as for the
// 64-place LDS tile; the event scale is the byte stride of a row of W2^T (an offset word is the row's byte offset, no
// multiply per event: ldw2 must be the minimal roundup(P, 256)); the arithmetic is integrate_chunk's.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_output_carry_l2(OutArgs a)
{
    const int lane = threadIdx.x;
    const int tile = blockIdx.x % a.ntiles, grp = blockIdx.x / a.ntiles;
    const int qs = (int)(((long long)grp * a.Q) / a.n_groups), qe = (int)(((long long)(grp + 1) * a.Q) / a.n_groups);
    const int place = tile * 64 + lane;                          // < ldw: the rows are zero padded to a multiple of 256
    const float *wb = a.w2t + (size_t)tile * 64;
    const uint32_t lane4 = (uint32_t)lane * 4u, voff = (uint32_t)(lane & 31) * 4u;
    const float thr = 0.99999994f;                               // 1 - 2^-24: the clip rounds it up to exactly 1.0
    float v = (place < a.P) ? a.vout[(size_t)grp * a.ldv + place] : 0.0f;
    for (int q = qs; q < qe; ++q) {
        const uint32_t *ev = a.events + (size_t)q * a.cap;
        const uint32_t *nwp = a.nwords + q;
        float cnt, acc, t, t2;
        uint32_t va, w0, w1, w2, w3, nch, soff, sl, sn, sc, so, cw, m, k;
        asm volatile(
#include "carry_l2_loop.inc"
            : [v0] "+&v"(v), [c0] "=&v"(cnt), [a0] "=&v"(acc), [t] "=&v"(t), [t2] "=&v"(t2), [va] "=&v"(va), [w0] "=&v"(w0),
              [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [nch] "=&s"(nch), [soff] "=&s"(soff), [sl] "=&s"(sl),
              [sn] "=&s"(sn), [sc] "=&s"(sc), [so] "=&s"(so), [cw] "=&s"(cw), [m] "=&s"(m), [k] "=&s"(k)
            : [ev] "s"(ev), [nwp] "s"(nwp), [wb] "s"(wb), [lane4] "v"(lane4), [voff] "v"(voff), [thr] "v"(thr)
            : "memory", "vcc", "scc",
              "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
              "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
              "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
              "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
              "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
              "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
              "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
              "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
              "v124", "v125", "v126", "v127");
        if (place < a.P) a.counts[(size_t)q * a.ldo + place] = cnt;
    }
    if (place < a.P) a.vout[(size_t)grp * a.ldv + place] = v;
}

// ------------------------------------------------------------------------------------------
// a5-a6, DENSE variant: the "descriptor x database GEMM" on the f32 matrix cores, for comparison with the
// event-driven kernel above.  v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf
// chain, so accumulating the dense [32 queries x F] spike tile of one time step against [F x 32 places]
// gives exactly the oracle's ascending-f chain (zero spikes contribute fmaf(0, w, acc) == acc).
//   workgroup = 4 wavefronts = 32 queries x 128 places; each wavefront keeps its 32 places' weights in
//   registers (one VGPR per k-pair), the step's spike tile is staged in LDS, membrane potentials and counters
//   of the 32x32 tile live in the 16 accumulator-layout registers; T steps inside the kernel.
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KP>   // k-pairs held in registers: F <= 2 * KP
__global__ void __launch_bounds__(256) k_output_counts_mfma(const float *__restrict__ spk, int Q, int T, int F, int lds_f,
                                                            const float *__restrict__ w2t, int ldw, int P,
                                                            float *__restrict__ counts, int ldo)
{
    constexpr int FP = 2 * KP + 1;                   // odd row stride: conflict-free column reads
    __shared__ float sA[2][32 * FP];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q0 = blockIdx.y * 32;
    const int p0 = blockIdx.x * 128 + wave * 32;
    const int col = lane & 31, hi = lane >> 5;
    float b[KP];                                     // B[k = 2m + hi][j = col]
#pragma unroll
    for (int m = 0; m < KP; ++m) {
        const int f = 2 * m + hi;
        b[m] = (f < F) ? w2t[(size_t)f * ldw + p0 + col] : 0.0f;   // w2t is zero padded to ldw >= roundup(P, 256)
    }
    f32x16 v, cnt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { v[r] = 0.0f; cnt[r] = 0.0f; }

    auto stage = [&](int t, int buf) {               // spike tile of step t: 32 queries x F -> LDS (zero padded)
        for (int i = threadIdx.x; i < 32 * 2 * KP; i += blockDim.x) {
            const int row = i / (2 * KP), f = i - row * (2 * KP);
            const int q = q0 + row;
            sA[buf][row * FP + f] = (q < Q && f < F) ? spk[((size_t)q * T + t) * lds_f + f] : 0.0f;
        }
    };
    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (t + 1 < T) stage(t + 1, buf ^ 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float *arow = &sA[buf][col * FP + hi];  // A[i = col][k = 2m + hi]
#pragma unroll
        for (int m = 0; m < KP; ++m)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * m], b[m], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float vv = v[r];
            cnt[r] += iaf_update(vv, acc[r]);
            v[r] = vv;
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 tile: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int p = p0 + col;
        if (q < Q && p < P) counts[(size_t)q * ldo + p] = cnt[r];
    }
}

// ------------------------------------------------------------------------------------------
// a7. sequence matcher, full matrix (query-major).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_seqmatch(const float *__restrict__ counts, int Ps, int ldo, int L,
                                                  float *__restrict__ S, int lds)
{
    const int i = blockIdx.y;
    const float fl = (float)L;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < Ps; j += gridDim.x * blockDim.x) {
        float n = 0.0f;
        for (int d = 0; d < L; ++d) n += counts[(size_t)(i + d) * ldo + j + d];   // exact: small integers
        S[(size_t)i * lds + j] = n / fl;
    }
}

// ------------------------------------------------------------------------------------------
// Wave-level sorted list, one entry per lane (lane 0 = best), ordered by (n desc, j desc).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool beats(float n1, int j1, float n2, int j2)
{
    return n1 > n2 || (n1 == n2 && j1 > j2);
}

// value of lane (l - 1), by DPP wave_shr:1 (no LDS crossbar traffic); lane 0 keeps its own value
__device__ __forceinline__ int lane_above(int x)
{
    return __builtin_amdgcn_update_dpp(x, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// wave-uniform lane index -> v_readlane_b32
__device__ __forceinline__ float bcast(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ int bcast(int x, int l) { return __builtin_amdgcn_readlane(x, l); }

__device__ __forceinline__ void wave_insert(float &ln, int &lj, float cn, int cj, int lane)
{
    const uint64_t m = __ballot(beats(cn, cj, ln, lj));   // the beaten entries form a suffix
    if (m == 0) return;
    const int pos = __builtin_ctzll(m);
    const float un = __int_as_float(lane_above(__float_as_int(ln)));
    const int uj = lane_above(lj);
    if (lane > pos) { ln = un; lj = uj; }
    else if (lane == pos) { ln = cn; lj = cj; }
}

// a7 + top-K.  One WAVEFRONT per query column (four per workgroup, no cross-wave merge): it scans 64-place
// chunks from the highest place index down (so that later ties never displace earlier ones), U chunks of loads
// in flight, and keeps a lane-distributed sorted top-64 list.
// LT = compile-time sequence length (0: use the run-time L), so that the L loads of a chunk are independent.
template <int LT>
__global__ void __launch_bounds__(256) k_seqmatch_topk(const float *__restrict__ counts, int ldo, int n_places,
                                                       int Lrt, int K, int place_offset, int n_queries,
                                                       int32_t *__restrict__ tk_idx, float *__restrict__ tk_val)
{
    const int L = LT ? LT : Lrt;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (i >= n_queries) return;
    float ln = -1.0f;   // empty entry: loses against every real candidate (n >= 0)
    int lj = -1;
    float tn = -1.0f;   // current K-th best
    int tj = -1;
    const int nchunks = (n_places + 63) >> 6;
    constexpr int U = 8;   // chunks (loads) in flight per wavefront: the scan is HBM-bound, not latency-bound
    for (int c0 = 0; c0 < nchunks; c0 += U) {
        float n[U];
        int j[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u;
            j[u] = (c < nchunks) ? n_places - 1 - (c * 64 + lane) : -1;
            const int jc = j[u] < 0 ? 0 : j[u];   // unconditional loads (clamped index), masked by j[u] below
            n[u] = 0.0f;
            if constexpr (LT > 0) {
#pragma unroll
                for (int d = 0; d < LT; ++d) n[u] += counts[(size_t)(i + d) * ldo + jc + d];
            } else {
                for (int d = 0; d < L; ++d) n[u] += counts[(size_t)(i + d) * ldo + jc + d];
            }
            // candidates are ranked on S = n / L like the reference's argsort, not on n: for sums beyond 2^22 two
            // different n can round to the same S and must then tie (L = 2, 4: an exact multiply; L = 1: nothing)
            n[u] = n[u] / (float)L;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u == 0 && c0 == 0) {
                // The list is empty: instead of 64 serial insertions, rank the first chunk by counting (position =
                // number of candidates that beat this one) and scatter it into place.  Lanes without a candidate
                // (n_places < 64) carry (-1, -1), lose to every real one and keep their lane order among themselves.
                const float n0 = (j[0] >= 0) ? n[0] : -1.0f;
                int rank = 0;
                for (int l = 0; l < 64; ++l) {
                    const float cn = bcast(n0, l);
                    const int cj = bcast(j[0], l);
                    rank += (beats(cn, cj, n0, j[0]) || (cj < 0 && j[0] < 0 && l < lane)) ? 1 : 0;
                }
                ln = __int_as_float(__builtin_amdgcn_ds_permute(rank << 2, __float_as_int(n0)));
                lj = __builtin_amdgcn_ds_permute(rank << 2, j[0]);
                tn = bcast(ln, K - 1);
                tj = bcast(lj, K - 1);
                continue;
            }
            uint64_t m = __ballot(j[u] >= 0 && beats(n[u], j[u], tn, tj));
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const float cn = bcast(n[u], b);
                const int cj = bcast(j[u], b);
                if (beats(cn, cj, tn, tj)) {
                    wave_insert(ln, lj, cn, cj, lane);
                    tn = bcast(ln, K - 1);
                    tj = bcast(lj, K - 1);
                }
            }
        }
    }
    if (lane < K) {
        tk_idx[(size_t)i * K + lane] = (lj >= 0) ? lj + place_offset : -1;
        tk_val[(size_t)i * K + lane] = (lj >= 0) ? ln : -INFINITY;
    }
}

// Merge of R per-shard lists: one wavefront per query column; values are compared as stored
// (they are n/L with one common L, a monotone map of n).
__global__ void __launch_bounds__(256) k_topk_merge(const int32_t *__restrict__ idx_in, const float *__restrict__ val_in,
                                                    int R, int Qs, int K,
                                                    int32_t *__restrict__ idx_out, float *__restrict__ val_out)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= Qs) return;
    float ln = -INFINITY;
    int lj = -1;
    for (int r = 0; r < R; ++r) {
        const size_t base = ((size_t)r * Qs + i) * K;
        for (int e = 0; e < K; ++e) {
            const int cj = idx_in[base + e];
            if (cj < 0) break;
            wave_insert(ln, lj, val_in[base + e], cj, lane);
        }
    }
    if (lane < K) {
        idx_out[(size_t)i * K + lane] = lj;
        val_out[(size_t)i * K + lane] = (lj >= 0) ? ln : -INFINITY;
    }
}


// ------------------------------------------------------------------------------------------
// f1. Training (`--train_model`): lens/train_model.py:157-177 + calc_stdp, lens/src/blitnet.py:183-254.
//   Every output neuron of a layer evolves independently given the sample sequence (its forward value,
//   STDP row, threshold and homeostasis only involve its own weight row), so 1, 2, 4, 8 or 16 adjacent lanes own one
//   row, keep it in registers for all T steps, and the per-step input vector is staged through LDS.  Per-element sign
//   masks are kept as packed bits.  IN_PAD: padded fan-in; x_table / mpre_table rows are zero padded to ldx >= IN_PAD.
// ------------------------------------------------------------------------------------------
struct TrainArgs {
    float *w; int ldw;
    float *thr;
    const uint8_t *exc; const uint8_t *inh; int ldm;
    const float *fire_rate; int have_rate;
    const float *x_table; const float *mpre_table; int ldx;
    const int32_t *order; const int32_t *force_idx;
    const float *eta_ip; const float *eta_stdp;
    int n_out; int n_in; int T;
};

// Stage one table row (IN_PAD floats, zero padded) into this wavefront's LDS buffer: 16-byte loads, lane l
// moves elements 4l..4l+3 of every 256-element piece.
template <int IN_PAD>
__device__ __forceinline__ void stage_row(const float *row, float *lds, int lane)
{
#pragma unroll
    for (int base = 0; base < IN_PAD; base += 256) {
        const int c = base + lane * 4;
        if (c < IN_PAD) *reinterpret_cast<float4 *>(lds + c) = *reinterpret_cast<const float4 *>(row + c);
    }
}

// value of the lane to the left (lane - 1) via DPP row_shr:1 (rows of 16 lanes; the first lane of a row gets 0)
__device__ __forceinline__ float lane_left(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111 /* row_shr:1 */, 0xf, 0xf, false));
}

// A wave-uniform value in an SGPR, whatever the compiler would have done with it (a plain `"s"` operand of a uniform
// select can still arrive in a VGPR).  The s_nop covers the wait state gfx950 needs between a VALU write of a VGPR
// and a v_readfirstlane of it: the compiler's hazard recognizer does not look inside an asm statement, and the
// producer of `x` is often the instruction right before it.
__device__ __forceinline__ uint32_t to_sgpr(uint32_t x)
{
    uint32_t r;
    asm volatile("s_nop 1\n v_readfirstlane_b32 %0, %1\n s_nop 1" : "=s"(r) : "v"(x));
    return r;
}

// The per-weight sections of a training step, four weights of a row per block of instructions.  Written out because
// the compiler turns the bit-mask blends into compare + select per element and keeps hundreds of per-element masks
// alive (512 registers, spills, one wavefront per SIMD).  With the weights touched only by these blocks the compiler
// keeps one copy of the row in registers across the wave-uniform branches around the sections.  B = bit of the
// block's first weight in the packed mask words.
//
// This step's inhibitory clamp (blitnet.py:235; UPDATE = false) and the homeostatic move (:246-251; UPDATE = true,
// run only on steps with eta != 0), per weight:
//   ci = clamp(w, -10, -1e-6);  wn = inhibitory ? ci : w;     then     w = wn + ((y * min(wn, 0)) * eta) * 50.
// Two blocks rather than one with both outcomes: a branch with weights written on both sides makes the compiler
// carry two copies of the row again.
template <int B, bool UPDATE>
__device__ __forceinline__ void homeostasis4(float &w0, float &w1, float &w2, float &w3, uint32_t ib, float y, float eta,
                                             float lo, float hi)
{
    float c0, c1, c2, c3;
    if constexpr (!UPDATE) {         // this step's inhibitory clamp
        uint32_t m0, m1, m2, m3;
        asm volatile(
            "v_med3_f32 %[c0], %[w0], %[lo], %[hi]\n v_med3_f32 %[c1], %[w1], %[lo], %[hi]\n"
            "v_med3_f32 %[c2], %[w2], %[lo], %[hi]\n v_med3_f32 %[c3], %[w3], %[lo], %[hi]\n"
            "v_bfe_i32 %[m0], %[ib], %[b0], 1\n v_bfe_i32 %[m1], %[ib], %[b0]+1, 1\n"
            "v_bfe_i32 %[m2], %[ib], %[b0]+2, 1\n v_bfe_i32 %[m3], %[ib], %[b0]+3, 1\n"
            "v_bfi_b32 %[w0], %[m0], %[c0], %[w0]\n v_bfi_b32 %[w1], %[m1], %[c1], %[w1]\n"
            "v_bfi_b32 %[w2], %[m2], %[c2], %[w2]\n v_bfi_b32 %[w3], %[m3], %[c3], %[w3]\n"
            : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2),
              [c3] "=&v"(c3), [m0] "=&v"(m0), [m1] "=&v"(m1), [m2] "=&v"(m2), [m3] "=&v"(m3)
            : [ib] "v"(ib), [lo] "v"(lo), [hi] "v"(hi), [b0] "n"(B));
    } else {                         // the homeostatic move of the (clamped) weights
        asm volatile(
            "v_min_f32 %[c0], 0, %[w0]\n v_min_f32 %[c1], 0, %[w1]\n v_min_f32 %[c2], 0, %[w2]\n v_min_f32 %[c3], 0, %[w3]\n"
            "v_mul_f32 %[c0], %[y], %[c0]\n v_mul_f32 %[c1], %[y], %[c1]\n v_mul_f32 %[c2], %[y], %[c2]\n v_mul_f32 %[c3], %[y], %[c3]\n"
            "v_mul_f32 %[c0], %[c0], %[eta]\n v_mul_f32 %[c1], %[c1], %[eta]\n v_mul_f32 %[c2], %[c2], %[eta]\n v_mul_f32 %[c3], %[c3], %[eta]\n"
            "v_mul_f32 %[c0], 0x42480000, %[c0]\n v_mul_f32 %[c1], 0x42480000, %[c1]\n"
            "v_mul_f32 %[c2], 0x42480000, %[c2]\n v_mul_f32 %[c3], 0x42480000, %[c3]\n"
            "v_add_f32 %[w0], %[w0], %[c0]\n v_add_f32 %[w1], %[w1], %[c1]\n v_add_f32 %[w2], %[w2], %[c2]\n v_add_f32 %[w3], %[w3], %[c3]\n"
            : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2),
              [c3] "=&v"(c3)
            : [y] "v"(y), [eta] "v"(eta));
    }
}

// One weight of the spike-forced STDP update (blitnet.py:185-215):  dw = (mpre * xd) * eta on excitatory and
// inhibitory weights (the reference's two sign flips cancel to the same bits), then the excitatory clamp.
#define LENS_STDP_FORCED(i)                                                                                   \
    "v_mul_f32 %[d], %[p" #i "], %[xd]\n v_bfe_i32 %[e], %[eb], %[b0]+" #i ", 1\n"                                 \
    "v_mul_f32 %[d], %[d], %[eta]\n v_bfe_i32 %[m], %[ib], %[b0]+" #i ", 1\n"                                     \
    "v_or_b32 %[m], %[m], %[e]\n v_and_b32 %[d], %[d], %[m]\n"                                                  \
    "v_add_f32 %[w" #i "], %[w" #i "], %[d]\n v_med3_f32 %[d], %[w" #i "], %[lo], %[hi]\n"                            \
    "v_bfi_b32 %[w" #i "], %[e], %[d], %[w" #i "]\n"
// First step of a call: every excitatory weight into its range 1e-6 .. 10 (blitnet.py:234).
#define LENS_CLAMP_EXC(i)                                                                                     \
    "v_med3_f32 %[d], %[w" #i "], %[lo], %[hi]\n v_bfe_i32 %[e], %[eb], %[b0]+" #i ", 1\n"                          \
    "v_bfi_b32 %[w" #i "], %[e], %[d], %[w" #i "]\n"
// The two sections that run on few steps only, in one block: the spike-forced STDP update (flag `forced`: some lane
// of this wavefront owns the forced row; its presynaptic row is read from LDS at `row_lds` + 4*K) and the
// first-step clamp (flag `first`); the caller skips the block when neither is set.  One weight after the other: speed
// does not matter here.
template <int B, int K>
__device__ __forceinline__ void rare_sections4(float &w0, float &w1, float &w2, float &w3, uint32_t eb, uint32_t ib, uint32_t row_lds,
                                               float xd, float eta, uint32_t forced, uint32_t first, float lo, float hi)
{
    float d, p0, p1, p2, p3;
    uint32_t m, e;
    asm volatile(
        "s_cmp_eq_u32 %[forced], 0\n s_cbranch_scc1 LENS_R1_%=\n"
        "ds_read_b32 %[p0], %[row] offset:%[koff]\n ds_read_b32 %[p1], %[row] offset:%[koff]+4\n"
        "ds_read_b32 %[p2], %[row] offset:%[koff]+8\n ds_read_b32 %[p3], %[row] offset:%[koff]+12\n s_waitcnt lgkmcnt(0)\n"
        LENS_STDP_FORCED(0) LENS_STDP_FORCED(1) LENS_STDP_FORCED(2) LENS_STDP_FORCED(3)
        "LENS_R1_%=:\n"
        "s_cmp_eq_u32 %[first], 0\n s_cbranch_scc1 LENS_R2_%=\n"
        LENS_CLAMP_EXC(0) LENS_CLAMP_EXC(1) LENS_CLAMP_EXC(2) LENS_CLAMP_EXC(3)
        "LENS_R2_%=:\n"
        : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [d] "=&v"(d), [m] "=&v"(m), [e] "=&v"(e),
          [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3)
        : [eb] "v"(eb), [ib] "v"(ib), [row] "v"(row_lds), [xd] "v"(xd), [eta] "v"(eta), [forced] "s"(forced),
          [first] "s"(first), [lo] "v"(lo), [hi] "v"(hi), [b0] "n"(B), [koff] "n"(4 * K)
        : "scc", "memory");
}
#undef LENS_STDP_FORCED
#undef LENS_CLAMP_EXC

// One weight of the free STDP update (blitnet.py:218-231): where the presynaptic value is positive, excitatory
// weights move by de, inhibitory ones by nde (both already zeroed when the neuron did not fire), then the
// excitatory clamp on the weights that moved -- on the first step of a call on every excitatory weight (flag
// `first`, an all-ones or all-zeros word ORed into the clamp's condition).  The compare's lane mask is consumed
// two instructions later (the wait states the hardware asks for between a VALU write of VCC and its use as a
// select mask).
#define LENS_STDP_FREE(i, t)                                                                                  \
    "v_cmp_lt_f32 vcc, 0, %[p" #i "]\n v_bfe_i32 %[e" #t "], %[eb], %[b0]+" #i ", 1\n"                               \
    "v_bfe_i32 %[m" #t "], %[ib], %[b0]+" #i ", 1\n v_cndmask_b32_e64 %[d" #t "], 0, -1, vcc\n"                      \
    "v_and_b32 %[m" #t "], %[m" #t "], %[nde]\n v_and_b32 %[c" #t "], %[e" #t "], %[d" #t "]\n"                        \
    "v_and_or_b32 %[m" #t "], %[c" #t "], %[de], %[m" #t "]\n v_and_b32 %[m" #t "], %[m" #t "], %[d" #t "]\n"            \
    "v_add_f32 %[w" #i "], %[w" #i "], %[m" #t "]\n v_and_or_b32 %[c" #t "], %[e" #t "], %[first], %[c" #t "]\n"        \
    "v_med3_f32 %[d" #t "], %[w" #i "], %[lo], %[hi]\n v_bfi_b32 %[w" #i "], %[c" #t "], %[d" #t "], %[w" #i "]\n"
template <int B>
__device__ __forceinline__ void stdp_free4(float &w0, float &w1, float &w2, float &w3, uint32_t eb, uint32_t ib,
                                           float p0, float p1, float p2, float p3, uint32_t de, uint32_t nde, uint32_t first,
                                           float lo, float hi)
{
    uint32_t d0, d1, m0, m1, e0, e1, c0, c1;       // two weights at a time: eight temporaries instead of sixteen
    asm volatile(LENS_STDP_FREE(0, 0) LENS_STDP_FREE(1, 1) LENS_STDP_FREE(2, 0) LENS_STDP_FREE(3, 1)
                 : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [d0] "=&v"(d0), [d1] "=&v"(d1), [m0] "=&v"(m0),
                   [m1] "=&v"(m1), [e0] "=&v"(e0), [e1] "=&v"(e1), [c0] "=&v"(c0), [c1] "=&v"(c1)
                 : [eb] "v"(eb), [ib] "v"(ib), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [de] "v"(de),
                   [nde] "v"(nde), [first] "v"(first), [lo] "v"(lo), [hi] "v"(hi), [b0] "n"(B)
                 : "vcc");
}
#undef LENS_STDP_FREE

// The sections over the HP weights of a lane (HP is a multiple of 4; a group of four never straddles a mask word).
template <int HP, int NB, bool UPDATE, int K = 0>
__device__ __forceinline__ void homeostasis_row(float (&w)[HP], const uint32_t (&ib)[NB], float y, float eta, float lo, float hi)
{
    if constexpr (K < HP) {
        homeostasis4<(K & 31), UPDATE>(w[K], w[K + 1], w[K + 2], w[K + 3], ib[K >> 5], y, eta, lo, hi);
        homeostasis_row<HP, NB, UPDATE, K + 4>(w, ib, y, eta, lo, hi);
    }
}
template <int HP, int NB, int K = 0>
__device__ __forceinline__ void rare_sections_row(float (&w)[HP], const uint32_t (&eb)[NB], const uint32_t (&ib)[NB], uint32_t row_lds,
                                                  float xd, float eta, uint32_t forced, uint32_t first, float lo, float hi)
{
    if constexpr (K < HP) {
        rare_sections4<(K & 31), K>(w[K], w[K + 1], w[K + 2], w[K + 3], eb[K >> 5], ib[K >> 5], row_lds, xd, eta, forced, first, lo, hi);
        rare_sections_row<HP, NB, K + 4>(w, eb, ib, row_lds, xd, eta, forced, first, lo, hi);
    }
}
template <int HP, int NB, int K = 0>
__device__ __forceinline__ void stdp_free_row(float (&w)[HP], const uint32_t (&eb)[NB], const uint32_t (&ib)[NB], const float *row,
                                              uint32_t de, uint32_t nde, uint32_t first, float lo, float hi)
{
    if constexpr (K < HP) {
        const float4 p = *reinterpret_cast<const float4 *>(row + K);
        stdp_free4<(K & 31)>(w[K], w[K + 1], w[K + 2], w[K + 3], eb[K >> 5], ib[K >> 5], p.x, p.y, p.z, p.w, de, nde, first, lo, hi);
        stdp_free_row<HP, NB, K + 4>(w, eb, ib, row, de, nde, first, lo, hi);
    }
}

// LPR = lanes per weight row (1, 2, 4, 8 or 16: at most one DPP row of lanes).  A row of more than 128 weights would not fit the 256 architectural
// VGPRs of one lane, so LPR adjacent lanes share it: lane LPR*r + h holds columns [h*HP, (h+1)*HP) of row r.  The
// forward chain runs over the first lane's columns, is handed to the next lane (DPP row_shr:1) and continues
// there, and so on: still ONE ascending fmaf chain per row.  The last lane's result is then broadcast to the row's
// lanes, every one of which updates its own columns.
template <int IN_PAD, int LPR, bool SPK>
__device__ __forceinline__ void train_layer_body(const TrainArgs &a)
{
    static_assert(IN_PAD % (4 * LPR) == 0 && IN_PAD <= 2048 && (LPR == 1 || LPR == 2 || LPR == 4 || LPR == 8 || LPR == 16),
                  "rows are staged in 256-element pieces and shared by a power-of-two group of lanes inside a DPP row");
    constexpr int HP = IN_PAD / LPR;              // columns per lane
    constexpr int NB = (HP + 31) / 32;
    constexpr int RPW = 64 / LPR;                 // rows per wavefront
    __shared__ __attribute__((aligned(16))) float s_row[IN_PAD];   // the step's input vector, read back as broadcasts
    const int lane = threadIdx.x;
    const int h = lane & (LPR - 1);
    const int o = blockIdx.x * RPW + lane / LPR;
    const bool valid = o < a.n_out;
    const int col0 = h * HP;
    float w[HP];
    uint32_t eb[NB], ib[NB];
    {   // rows are zero padded to ldw / ldm >= IN_PAD by the caller: unpredicated loads, row index clamped
        const size_t ro = (size_t)(valid ? o : 0);
        const float *wr = a.w + ro * a.ldw + col0;
        const uint8_t *er = a.exc + ro * a.ldm + col0, *ir = a.inh + ro * a.ldm + col0;
#pragma unroll
        for (int k4 = 0; k4 < HP; k4 += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(wr + k4);
            w[k4] = v.x; w[k4 + 1] = v.y; w[k4 + 2] = v.z; w[k4 + 3] = v.w;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            uint32_t e = 0u, i = 0u;
            const int lim = (HP - 32 * b) < 32 ? (HP - 32 * b) : 32;
#pragma unroll 1
            for (int j = 0; j < lim; ++j) {
                e |= (er[32 * b + j] ? 1u : 0u) << j;
                i |= (ir[32 * b + j] ? 1u : 0u) << j;
            }
            eb[b] = valid ? e : 0u;
            ib[b] = valid ? i : 0u;
        }
        // lanes beyond the last row train a copy of row 0 and store nothing
    }
    float thr = valid ? a.thr[o] : 0.0f;
    const float fr = valid ? a.fire_rate[o] : 0.0f;
    const float *my_row = s_row + col0;               // this lane's segment of the staged input vector
    using lds_f = __attribute__((address_space(3))) float *;
    const uint32_t row_lds = (uint32_t)(uintptr_t)(lds_f)my_row;

    for (int t = 0; t < a.T; ++t) {
        const int s = a.order[t];
        const float eta = a.eta_stdp[t], eip = a.eta_ip[t];
        stage_row<IN_PAD>(a.x_table + (size_t)s * a.ldx, s_row, lane);
        float acc = 0.0f;                                    // y = layer.w(x): ascending fmaf chain
#pragma unroll
        for (int ph = 0; ph < LPR; ++ph) {
            if (ph >= 1) {                                           // lane h = ph continues the chain of lane ph - 1
                const float left = lane_left(acc);                   // DPP with all lanes active, then select
                acc = (h == ph) ? left : acc;
            }
#pragma unroll
            for (int k4 = 0; k4 < HP; k4 += 4) {
                if ((k4 & 15) == 0) __builtin_amdgcn_sched_barrier(0);   // bound live ranges of the unrolled body
                const float4 p = *reinterpret_cast<const float4 *>(s_row + ph * HP + k4);   // wave-uniform: broadcast
                float c = acc;
                c = __builtin_fmaf(p.x, w[k4], c);
                c = __builtin_fmaf(p.y, w[k4 + 1], c);
                c = __builtin_fmaf(p.z, w[k4 + 2], c);
                c = __builtin_fmaf(p.w, w[k4 + 3], c);
                acc = (LPR == 1 || h == ph) ? c : acc;       // only the half that owns this segment advances
            }
        }
        if (LPR > 1)                                         // every lane of the row holds y: the last lane's value
            acc = __shfl(acc, lane | (LPR - 1), 64);
        const float y = acc;                                 // spikes_noclp
        float post = y - thr;                                // clamp_spikes
        post = __builtin_fminf(__builtin_fmaxf(post, 0.0f), 1.0f);
        float lx = 0.0f;                                     // layer.x
        // Weight clamps (blitnet.py:234-235) are no-ops on elements that are already in range.  Excitatory
        // weights only leave their range through this step's STDP update (homeostasis adds +-0 to positive
        // weights), so they are clamped where they are updated -- and all of them at a call's first step, where
        // the initial weights may be below 1e-6.  Inhibitory weights are moved by homeostasis every step.
        const uint32_t first = to_sgpr(t == 0 ? 1u : 0u);   // flags of the blocks: SGPRs
        if constexpr (SPK) {                                 // blitnet.py:185-215
            lx = (o == a.force_idx[t]) ? 0.5f : 0.0f;
            float xd = lx - post;
            xd = __builtin_fminf(__builtin_fmaxf(xd, 0.0f), 1.0f);
            // only the wavefront that owns row idx has work here: it stages the presynaptic row over the input row
            const uint32_t forced = to_sgpr(__builtin_amdgcn_ballot_w64(xd > 0.0f) != 0 ? 1u : 0u);
            if (forced) stage_row<IN_PAD>(a.mpre_table + (size_t)s * a.ldx, s_row, lane);
            // exc: ((mpre*xd)*1)*eta ; inh: ((-mpre*xd)*1)*(eta*-1) -- the same bits
            if (forced | first) rare_sections_row<HP, NB>(w, eb, ib, row_lds, xd, eta, forced, first, 1e-06f, 10.0f);
        } else {                                             // blitnet.py:218-231
            const float d = 0.5f - post;
            const uint32_t b2 = (post > 0.0f) ? 0xffffffffu : 0u;
            const uint32_t de = __float_as_uint(d * eta) & b2, nde = __float_as_uint(d * (eta * -1.0f)) & b2;
            stdp_free_row<HP, NB>(w, eb, ib, my_row, de, nde, 0u - first, 1e-06f, 10.0f);
        }
        if (a.have_rate && eip > 0.0f) {                     // blitnet.py:239-243
            thr = thr + eip * (lx - fr);
            thr = (thr < 0.0f) ? 0.0f : thr;
        }
        // blitnet.py:246-251 (any(W) is implied: it adds 0 otherwise), with this step's inhibitory clamp (:235) first
        homeostasis_row<HP, NB, false>(w, ib, y, eta, -10.0f, -1e-06f);
        if (eta != 0.0f) homeostasis_row<HP, NB, true>(w, ib, y, eta, -10.0f, -1e-06f);
    }
    if (valid) {
        float *wr = a.w + (size_t)o * a.ldw + col0;
#pragma unroll
        for (int k4 = 0; k4 < HP; k4 += 4)
            *reinterpret_cast<float4 *>(wr + k4) = make_float4(w[k4], w[k4 + 1], w[k4 + 2], w[k4 + 3]);
        if (h == 0) a.thr[o] = thr;
    }
}

// One kernel per (padded fan-in, lanes per row) with its occupancy spelled out (the attribute takes a literal only):
// the number of wavefronts per SIMD whose registers (weights per lane + ~45 working registers) fit the 512 per lane.
#define LENS_TRAIN_KERNEL(P, LPR, WAVES)                                                                   \
    template <bool SPK>                                                                                   \
    __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k_train_layer_##P##_##LPR(TrainArgs a) \
    {                                                                                                     \
        train_layer_body<P, LPR, SPK>(a);                                                                 \
    }
LENS_TRAIN_KERNEL(64, 1, 5) LENS_TRAIN_KERNEL(128, 1, 3) LENS_TRAIN_KERNEL(128, 2, 5) LENS_TRAIN_KERNEL(200, 1, 2)
LENS_TRAIN_KERNEL(200, 2, 3) LENS_TRAIN_KERNEL(208, 1, 2)
LENS_TRAIN_KERNEL(208, 2, 3) LENS_TRAIN_KERNEL(208, 4, 5) LENS_TRAIN_KERNEL(256, 2, 3) LENS_TRAIN_KERNEL(256, 4, 5)
LENS_TRAIN_KERNEL(512, 4, 3) LENS_TRAIN_KERNEL(512, 8, 5) LENS_TRAIN_KERNEL(1024, 8, 3) LENS_TRAIN_KERNEL(2048, 16, 3)
#undef LENS_TRAIN_KERNEL

// y[s][o] = clamp(chain_n fmaf(x[s][n], w[o][n]) - thr[o], 0, 1)  (train_model.py:163-168)
// A workgroup owns 64 outputs x 4 samples.  The 64 x 64 pieces of w are staged through LDS (loaded row-wise, coalesced;
// read column-wise, the odd row stride keeps the banks apart); x[s][n] is one broadcast load per wavefront.
__global__ void __launch_bounds__(256) k_layer_forward_clamped(const float *__restrict__ x, int S, int ldx,
                                                               const float *__restrict__ w, int ldw,
                                                               const float *__restrict__ thr, int n_in, int n_out,
                                                               float *__restrict__ y, int ldy)
{
    __shared__ float sw[64][65];
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
    const int o0 = blockIdx.x * 64, o = o0 + lane;
    const int s = blockIdx.y * 4 + row;
    const float *xs = x + (size_t)(s < S ? s : 0) * ldx;
    float acc = 0.0f;
    for (int n0 = 0; n0 < n_in; n0 += 64) {
        __syncthreads();
        for (int r = row; r < 64; r += 4)
            sw[r][lane] = (o0 + r < n_out && n0 + lane < n_in) ? w[(size_t)(o0 + r) * ldw + n0 + lane] : 0.0f;
        __syncthreads();
        const int kn = (n_in - n0) < 64 ? (n_in - n0) : 64;
        for (int k = 0; k < kn; ++k) acc = __builtin_fmaf(xs[n0 + k], sw[lane][k], acc);
    }
    if (s >= S || o >= n_out) return;
    float v = acc - thr[o];
    v = __builtin_fminf(__builtin_fmaxf(v, 0.0f), 1.0f);
    y[(size_t)s * ldy + o] = v;
}

// ------------------------------------------------------------------------------------------
// f2. Sum-of-absolute-differences baseline (`--sad`): torch.cdist(query, reference, p=1) of lens/src/sad.py:38
// on the raw uint8 frames.  A workgroup stages QT query frames in LDS; every lane owns one reference frame,
// streams it 16 bytes at a time and accumulates |q - r| with v_sad_u8 (4 bytes per instruction) against the QT
// broadcast query words.  Integer sums are exact; the result is converted to float32 once.
// ------------------------------------------------------------------------------------------
constexpr int SAD_QT = 8;

__global__ void __launch_bounds__(256) k_sad_matrix(const uint8_t *__restrict__ q, int Q, const uint8_t *__restrict__ r,
                                                    int P, int HW, float *__restrict__ out, int ldo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int q0 = blockIdx.y * SAD_QT;
    const int nq = (Q - q0) < SAD_QT ? (Q - q0) : SAD_QT;
    const int hw16 = HW >> 4;                       // 16-byte words per frame (HW % 16 == 0 on this path)
    uint4 *sq = reinterpret_cast<uint4 *>(smem);     // [SAD_QT][hw16]
    for (int i = threadIdx.x; i < SAD_QT * hw16; i += blockDim.x) {
        const int qi = i / hw16, k = i - qi * hw16;
        sq[i] = (qi < nq) ? reinterpret_cast<const uint4 *>(q + (size_t)(q0 + qi) * HW)[k] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uint4 *rr = reinterpret_cast<const uint4 *>(r + (size_t)p * HW);
    uint32_t acc[SAD_QT];
#pragma unroll
    for (int qi = 0; qi < SAD_QT; ++qi) acc[qi] = 0u;
    for (int k = 0; k < hw16; ++k) {
        const uint4 rv = rr[k];
#pragma unroll
        for (int qi = 0; qi < SAD_QT; ++qi) {
            const uint4 qv = sq[qi * hw16 + k];      // wave-uniform address: broadcast
            acc[qi] = __builtin_amdgcn_sad_u8(qv.x, rv.x, acc[qi]);
            acc[qi] = __builtin_amdgcn_sad_u8(qv.y, rv.y, acc[qi]);
            acc[qi] = __builtin_amdgcn_sad_u8(qv.z, rv.z, acc[qi]);
            acc[qi] = __builtin_amdgcn_sad_u8(qv.w, rv.w, acc[qi]);
        }
    }
#pragma unroll
    for (int qi = 0; qi < SAD_QT; ++qi)
        if (qi < nq) out[(size_t)(q0 + qi) * ldo + p] = (float)acc[qi];
}

// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
constexpr int LDS_BUDGET = 160 * 1024;   // gfx950: 160 KiB per CU, one workgroup may take all of it
constexpr int NUM_CUS = 256;             // MI355X

template <typename K>
int set_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

template <int NPL, int FPL>
int launch_feature(const FeatArgs &a, hipStream_t st)
{
    constexpr int NP = NPL * 64, FP = FPL * 64;
    const size_t lds = (((size_t)a.T * NP + 15) & ~(size_t)15) + (size_t)a.N * FP * 4 + 64;   // tables + 16 work estimates
    const int wpb = feat_threads<NPL, FPL>() / 64;   // wavefronts per workgroup sharing the staged tables
    const int blocks = (a.n_streams + wpb - 1) / wpb;
    if (lds <= (size_t)LDS_BUDGET - 1024) {
        auto k = k_feature_events<NPL, FPL, true>;
        if constexpr (NPL == 2 && FPL == 4) {
            // The assembly step loop addresses the staged tables from LDS address 0: the kernel must not own any
            // static LDS (a __shared__ variable added to it would shift the dynamic block and silently break it).
            static std::atomic<int> lds_base_checked{0};
            if (!lds_base_checked.load(std::memory_order_acquire)) {
                hipFuncAttributes at;
                if (hipError_t e = hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k))) return (int)e;
                if (at.sharedSizeBytes != 0) return LENS_ERR_UNSUPPORTED;
                lds_base_checked.store(1, std::memory_order_release);
            }
        }
        if (int e = set_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(wpb * 64), lds, st, a);
    } else {
        hipLaunchKernelGGL((k_feature_events<NPL, FPL, false>), dim3(blocks), dim3(wpb * 64), 0, st, a);
    }
    return (int)hipGetLastError();
}

template <int NPL>
int dispatch_feature_f(const FeatArgs &a, hipStream_t st)
{
    const int fpl = (a.F + 63) / 64;
    if (fpl <= 1) return launch_feature<NPL, 1>(a, st);
    if (fpl <= 2) return launch_feature<NPL, 2>(a, st);
    if (fpl <= 4) return launch_feature<NPL, 4>(a, st);
    if (fpl <= 8) return launch_feature<NPL, 8>(a, st);
    return launch_feature<NPL, 32>(a, st);
}

template <int NPL, int FPL>
int launch_currents(const FeatArgs &a, float *cur, int ldc, hipStream_t st)
{
    constexpr int NP = NPL * 64, FP = FPL * 64;
    const size_t lds = (((size_t)a.T * NP + 15) & ~(size_t)15) + (size_t)a.N * FP * 4;
    const int wpb = feat_threads<NPL, FPL>() / 64;
    const int blocks = (a.Q + wpb - 1) / wpb;
    if (lds <= (size_t)LDS_BUDGET - 1024) {
        auto k = k_feature_currents<NPL, FPL, true>;
        if (int e = set_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(wpb * 64), lds, st, a, cur, ldc);
    } else {
        hipLaunchKernelGGL((k_feature_currents<NPL, FPL, false>), dim3(blocks), dim3(wpb * 64), 0, st, a, cur, ldc);
    }
    return (int)hipGetLastError();
}

template <int NPL>
int dispatch_currents_f(const FeatArgs &a, float *cur, int ldc, hipStream_t st)
{
    const int fpl = (a.F + 63) / 64;
    if (fpl <= 1) return launch_currents<NPL, 1>(a, cur, ldc, st);
    if (fpl <= 2) return launch_currents<NPL, 2>(a, cur, ldc, st);
    if (fpl <= 4) return launch_currents<NPL, 4>(a, cur, ldc, st);
    if (fpl <= 8) return launch_currents<NPL, 8>(a, cur, ldc, st);
    return launch_currents<NPL, 16>(a, cur, ldc, st);
}

// dynamic LDS of k_feature_fire: the per-step tables of one query (see the kernel)
size_t fire_lds_bytes(int T, int F, int cap_words)
{
    const size_t NW = (size_t)(F + 63) / 64;
    const size_t mixed_words = ((size_t)cap_words / EV_WORDS + 31) / 32;
    return 2 * (((size_t)T * NW * 12 + 7) & ~(size_t)7) + FIRE_HELPERS * ((2 * (size_t)T + 1) + mixed_words) * 4;
}

template <int PPL, bool W_LDS, bool LONE = false>
int launch_output(const OutArgs &a, int threads, int grid, hipStream_t st)
{
    // carried state: one wavefront per workgroup, no query ranking -- only the (unused) counter word behind the tile,
    // so that three 50 KB tiles still fit one CU
    const size_t lds = (W_LDS ? (size_t)a.F * 64 * PPL * 4 : 0) + (a.carry ? OUT_BOOK_CARRY_BYTES : OUT_BOOK_BYTES);
    auto k = k_output_counts<PPL, W_LDS, LONE>;
    if constexpr (PPL == 3) {
        // The assembly loop reads the third places relative to LDS address 0: the kernel must not own static LDS
        // (a __shared__ variable added to it would shift the dynamic block and silently break the addressing).
        static std::atomic<int> lds_base_checked{0};
        if (!lds_base_checked.load(std::memory_order_acquire)) {
            hipFuncAttributes at;
            if (hipError_t e = hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k))) return (int)e;
            if (at.sharedSizeBytes != 0) return LENS_ERR_UNSUPPORTED;
            lds_base_checked.store(1, std::memory_order_release);
        }
    }
    if (int e = set_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(threads), lds, st, a);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Library-owned spike-stream workspace of lens_forward_counts_static (the entry point for callers that cannot
// allocate).  Every GPU the library runs on holds its own copy of this array, at its own address, until the
// process ends; the host never uses the array's name as an address, only the per-device pointer that
// hipGetSymbolAddress returns.
// ------------------------------------------------------------------------------------------
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)LENS_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[MAX_DEVICES];   // device address of `scratch` on each device, resolved on first use

// Device address of the scratch array on the CURRENT device (0 = ok, else a status for the caller).
int static_scratch(unsigned char **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return LENS_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return LENS_ERR_UNSUPPORTED;
        g_scratch_ptr[dev] = static_cast<unsigned char *>(p);
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

}   // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int lens_amd_abi_version(void) { return LENS_AMD_ABI_VERSION; }

const char *lens_amd_strerror(int status)
{
    switch (status) {
    case LENS_OK: return "ok";
    case LENS_ERR_NULL: return "a required pointer is NULL";
    case LENS_ERR_SIZE: return "a size or stride argument is out of range";
    case LENS_ERR_UNSUPPORTED: return "request is outside what the kernels implement";
    case LENS_ERR_ALIGN: return "a pointer or leading dimension violates the required alignment";
    case LENS_ERR_SCRATCH: return "the batch needs more spike-stream workspace than the library's static scratch array holds";
    default: return "";
    }
}

int lens_accumulate_events(const int16_t *d_xy, const int64_t *d_frame_off, int n_frames, int roi,
                           uint8_t *d_frames, int64_t *d_first_bad, void *stream)
{
    if (!d_frame_off || !d_frames || !d_first_bad) return LENS_ERR_NULL;
    if (n_frames < 0 || roi <= 0) return LENS_ERR_SIZE;
    if (roi > 200) return LENS_ERR_UNSUPPORTED;
    if (n_frames == 0) return LENS_OK;
    if (!d_xy) return LENS_ERR_NULL;
    if ((uintptr_t)d_xy & 3) return LENS_ERR_ALIGN;
    const size_t lds = (size_t)roi * roi * 4;
    if (int e = set_lds(k_accumulate_events, lds)) return e;
    hipLaunchKernelGGL(k_accumulate_events, dim3(n_frames), dim3(1024), lds, (hipStream_t)stream, d_xy, d_frame_off,
                       roi, d_frames, reinterpret_cast<unsigned long long *>(d_first_bad));
    return (int)hipGetLastError();
}

int lens_build_threshold_codes(const float *h_u_sel, int T, int N, uint8_t *h_codes)
{
    if (!h_u_sel || !h_codes) return LENS_ERR_NULL;
    if (T < 0 || N < 0) return LENS_ERR_SIZE;
    float th[256];
    for (int p = 0; p < 256; ++p) th[p] = (float)p / 255.0f;   // dataset.py:23 (f32 true division)
    for (size_t i = 0; i < (size_t)T * N; ++i) {
        const float u = h_u_sel[i];
        int lo = 1, hi = 255;   // smallest p in [1,255] with u < th[p]; th is increasing, th[255] = 1 > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (u < th[mid]) hi = mid; else lo = mid + 1;
        }
        h_codes[i] = (uint8_t)lo;
    }
    return LENS_OK;
}

int lens_build_active_steps(const uint8_t *h_codes, int T, int N, uint16_t *h_active_steps)
{
    if (!h_codes || !h_active_steps) return LENS_ERR_NULL;
    if (T < 0 || T > 65535 || N < 0) return LENS_ERR_SIZE;
    for (int n = 0; n < N; ++n) {
        uint32_t hist[256] = {0};
        for (int t = 0; t < T; ++t) hist[h_codes[(size_t)t * N + n]]++;
        uint32_t run = 0;
        for (int p = 0; p < 256; ++p) {             // input n spikes at step t iff px >= code[t][n]
            run += hist[p];
            h_active_steps[(size_t)n * 256 + p] = (uint16_t)run;
        }
    }
    return LENS_OK;
}

size_t lens_event_capacity_words(int T, int F)
{
    const size_t per_step = (size_t)(F > 1 ? F : 1);   // events per step, 2 words each
    return ((2 * (size_t)(T > 0 ? T : 0) * per_step) + (EV_WORDS - 1)) & ~(size_t)(EV_WORDS - 1);
}

// Places per lane of the output kernel for F features and P places: among the tiles whose W2^T columns fit in LDS,
// the one with the least work.  0 = no tile fits (weights are read through L2).
// Batch mode (carry == 0): places after padding P to whole tiles x the measured relative time per place of that
// width (wider tiles spread the per-event scalar work and branches over more places; measured at 100 000 places:
// 1.79 / 1 / 0.87 / 0.69 for 64 / 128 / 192 / 256).  A database of a few tiles runs as a single round of
// workgroups, where the coarser (query, tile) items of the wide tiles balance worse: measured +7 % at 1000 places.
// Carried state (carry != 0) has three layouts, chosen by a cost model in microseconds per query of a stream (every
// stream is sequential in its query index, so what counts is the wall time of one query-step times the rounds of
// workgroups; all figures measured at F = 200):
//   LONE_LDS  one wavefront per (64-place tile, stream), a workgroup of its own with its own copy of the tile in LDS:
//             14.7 us per query, as long as the CUs' LDS surely holds all of them at once (one tile less than fits per
//             CU: three 50 KB tiles on EVERY CU is more than the dispatcher achieves; 768 tiles x 1 stream: 7.24 ms per
//             256 queries instead of 3.9).  64-place tiles: a lone wavefront takes 15.9 / 18.5 / 23.4 us on a 64- / 128- /
//             192-place tile, and the narrow one keeps the most tiles resident.
//   LONE_L2   the same wavefronts reading their rows through L2 (k_output_carry_l2), all resident whatever the database
//             size: 17-19 us per query up to ~1500 wavefronts, beyond that bound by L2 / Infinity Cache traffic
//             (12 501 places: 6272 wavefronts 46 us, 50 176: 314 us; 100 000 places: 25 008 wavefronts 321 us).
//   SHARED    the batch mode's wide LDS tile shared by up to 16 streams, one wavefront each (workgroup = tile x 16
//             streams): a query-step costs what a wavefront's query costs in batch mode with that many wavefronts per
//             SIMD (31.5 us at four on the 192-place tile), but 16 streams advance per tile staged -- the layout for
//             several streams on a large database (100 000 places x 16 streams, ms per 256 queries: 5.14 as lone
//             wavefronts through L2, 1.52 shared; x 4: 5.18 / 4.48; x 2: 5.51 / 8.56; 12 501 places x 32: 0.37 / 0.21;
//             1000 places x 16: 0.28 on lone LDS tiles / 0.35 shared -- scripts/experiments/carry_layouts.py).
//             The workgroups of a launch all take the same time, so a grid of a few workgroups more than whole rounds of
//             the CUs costs a whole round more (100 000 places x 16 streams: 521 tiles, 2.04 rounds).  Where it pays, the
//             wide tiles cover whole rounds only and the remaining places run as a second launch of lone wavefronts on
//             128-place LDS tiles (same event scale as the 128- and 192-place shared tiles, 18.5 us per query-step, one
//             100 KB tile per CU): OutPlan::main_tiles.  LENS_AMD_CARRY_TAIL=0 switches that off.
// LENS_AMD_CARRY_L2=0/1 and LENS_AMD_CARRY_SHARED=0/1/2/3/4 force a choice (tuning and tests; read at every call).
enum { OUT_BATCH = 0, OUT_LONE_LDS = 1, OUT_LONE_L2 = 2, OUT_SHARED = 3 };
struct OutPlan { int ppl; int mode; int main_tiles = 0; };   // main_tiles > 0 (SHARED only): that many wide tiles, the rest as a tail (below)

static int output_ppl_batch(int F, int P)
{
    static const struct { int ppl; double rel; } widths[] = {{4, 0.69}, {3, 0.87}, {2, 1.00}, {1, 1.79}};
    // tuning override: LENS_AMD_OUTPUT_PPL=1..4 forces that width wherever it fits (read once)
    static const int forced = [] { const char *e = getenv("LENS_AMD_OUTPUT_PPL"); return e ? atoi(e) : 0; }();
    if (forced >= 1 && forced <= 4 && (size_t)F * 64 * forced * 4 <= (size_t)LDS_BUDGET - 4096) return forced;
    int best = 0;
    double best_cost = 0.0;
    for (const auto &c : widths) {
        const size_t tile_bytes = (size_t)F * 64 * c.ppl * 4;
        if (tile_bytes > (size_t)LDS_BUDGET - 4096) continue;
        const int pt = 64 * c.ppl;
        const size_t tiles = ((size_t)(P > 0 ? P : 1) + pt - 1) / pt;
        const double cost = (double)tiles * pt * c.rel * (c.ppl >= 3 && tiles < 128 ? 1.07 : 1.0);
        if (!best || cost < best_cost) { best = c.ppl; best_cost = cost; }
    }
    return best;
}

static OutPlan output_plan(int F, int P, int carry, int n_groups)
{
    if (!carry) return {output_ppl_batch(F, P), OUT_BATCH};
    const size_t places = (size_t)(P > 0 ? P : 1), S = (size_t)(n_groups > 0 ? n_groups : 1);
    const size_t tile1 = (size_t)F * 64 * 4;
    if (tile1 > (size_t)LDS_BUDGET - 4096) return {0, OUT_LONE_LDS};     // no tile fits: 256 places per lone wavefront through L2
    // read at every call (unlike the other overrides), so that a test can walk one shape through every layout
    const char *env_l2 = getenv("LENS_AMD_CARRY_L2"), *env_shared = getenv("LENS_AMD_CARRY_SHARED");
    const int force_l2 = env_l2 ? atoi(env_l2) : -1;
    const int force_shared = env_shared ? atoi(env_shared) : -1;    // 0: never, 1: always (cheapest width), 2..4: always, that many places per lane
    const char *env_tail = getenv("LENS_AMD_CARRY_TAIL");
    const bool tail_allowed = !(env_tail && atoi(env_tail) == 0);
    // lone wavefronts on 64-place tiles
    const size_t per_cu = (size_t)LDS_BUDGET / (tile1 + 512);            // allocation granularity
    const size_t safe = NUM_CUS * (per_cu <= 2 ? 1 : per_cu > 33 ? 32 : per_cu - 1);
    const size_t waves = ((places + 63) / 64) * S;
    bool l2 = waves > safe;
    if (force_l2 == 0 || force_l2 == 1) l2 = force_l2 == 1;
    // through L2 an event's offset word is the byte offset of its row of W2^T (bit 31 is the chunk's flag)
    if ((unsigned long long)F * (((places + 255) / 256) * 256) * 4ull >= (1ull << 31)) l2 = false;
    OutPlan best{1, l2 ? OUT_LONE_L2 : OUT_LONE_LDS};
    // through L2 the wavefronts share the caches' bandwidth: 167 KB of rows per wavefront and query at the benchmark's
    // density, ~26 TB/s from L2 while the database fits the L2s, ~13 TB/s from the Infinity Cache beyond
    const double per_wave = (double)F * (double)(((places + 255) / 256) * 256) * 4.0 <= 24.0e6 ? 0.0066 : 0.0128;
    double best_cost = l2 ? (per_wave * (double)waves > 18.9 ? per_wave * (double)waves : 18.9)
                          : 14.7 * (double)((waves + safe - 1) / safe);
    if (force_shared == 0) return best;
    // up to 16 streams on a shared wide tile: lone = one wavefront alone on the tile, busy = per wavefront at four per SIMD
    // (busy: measured with 16 queries per stream and launch, the tile's staging included: 100 000 places x 16 streams 1.52 ms
    // per 256 queries = 3 rounds x 31.7 us; 12 501 x 16 on 128-place tiles 0.364 ms = 22.7 us, x 32: 26 us)
    static const struct { int ppl; double lone, busy; } shared[] = {{4, 28.0, 33.3}, {3, 23.4, 31.5}, {2, 18.5, 24.0}};
    const size_t wpw = S < 16 ? S : 16;                                   // wavefronts per workgroup
    for (const auto &c : shared) {
        if ((size_t)F * 64 * c.ppl * 4 > (size_t)LDS_BUDGET - 4096) continue;
        if (force_shared >= 2 && force_shared != c.ppl) continue;
        const size_t pt = 64 * (size_t)c.ppl;
        const size_t wgs = ((places + pt - 1) / pt) * ((S + 15) / 16);
        const double per_simd = (double)((wpw + 3) / 4);
        const double t = c.lone > c.busy * per_simd / 4.0 ? c.lone : c.busy * per_simd / 4.0;
        double cost = t * (double)((wgs + NUM_CUS - 1) / NUM_CUS);
        // whole rounds on the wide tiles + the remaining places as lone wavefronts on 128-place tiles (one per CU)
        int main_tiles = 0;
        const size_t sg = (S + 15) / 16;
        // wide tiles in whole rounds; an even number of 192-place tiles, so that the tail starts on a multiple of 128 places
        // (its last tile then ends within the padded row of W2^T)
        const size_t whole = ((wgs / NUM_CUS) * NUM_CUS / sg) & ~(size_t)(c.ppl == 3 ? 1 : 0);
        if (tail_allowed && c.ppl <= 3 && whole > 0 && whole * pt < places && (size_t)F * 512 + 512 <= (size_t)LDS_BUDGET) {
            const size_t tail_wgs = ((places - whole * pt + 127) / 128) * S;
            const double split = t * (double)((whole * sg + NUM_CUS - 1) / NUM_CUS) + 18.5 * (double)((tail_wgs + NUM_CUS - 1) / NUM_CUS);
            if (split < 0.95 * cost) { cost = split; main_tiles = (int)whole; }
        }
        // (a predicted gain below 10 % is within the model's error: the lone wavefronts stay)
        if (force_shared >= 1 ? (best.mode != OUT_SHARED || cost < best_cost) : cost < (best.mode == OUT_SHARED ? 1.0 : 0.9) * best_cost) {
            best = {c.ppl, OUT_SHARED, main_tiles};
            best_cost = cost;
        }
    }
    return best;
}

static int output_ppl(int F, int P, int carry, int n_groups) { return output_plan(F, P, carry, n_groups).ppl; }

int lens_output_tile_places(int F, int P, int carry, int n_groups)
{
    return 64 * output_ppl(F, P, carry, n_groups);
}

int lens_output_layout(int F, int P, int carry, int n_groups)
{
    return output_plan(F, P, carry, n_groups).mode;
}

int lens_output_tail_places(int F, int P, int carry, int n_groups)
{
    const OutPlan pl = output_plan(F, P, carry, n_groups);
    return pl.mode == OUT_SHARED && pl.main_tiles ? P - pl.main_tiles * 64 * pl.ppl : 0;
}

int lens_output_reads_through_l2(int F, int P, int carry, int n_groups)
{
    const OutPlan pl = output_plan(F, P, carry, n_groups);
    return pl.ppl == 0 || pl.mode == OUT_LONE_L2;
}

int lens_event_scale(int F, int P, int carry, int n_groups)
{
    const OutPlan pl = output_plan(F, P, carry, n_groups);
    const int ppl = pl.ppl;
    if (pl.mode == OUT_LONE_L2) return 4 * (((P > 0 ? P : 1) + 255) / 256 * 256);   // bytes of a row of W2^T at the minimal ldw2
    if (ppl == 3) return 512;   // the 128 places read as pairs; the 64 third places sit in a plane of half that stride
    return ppl ? 64 * ppl * 4 : 1;
}

int lens_feature_events(const uint8_t *d_frames, int Q, int HW, const int32_t *d_sel, int N,
                        const uint8_t *d_codes, int T, const float *d_w1t, int F, int ldw1,
                        float *d_vfeat, int n_streams, int carry,
                        uint32_t *d_events, int cap_words, uint32_t *d_nwords, int ev_scale,
                        float *d_featspk, const uint16_t *d_active_steps, void *stream)
{
    if (!d_frames || !d_sel || !d_codes || !d_w1t || !d_events || !d_nwords) return LENS_ERR_NULL;
    if (carry && !d_vfeat) return LENS_ERR_NULL;
    if (Q < 0 || HW <= 0 || N <= 0 || T <= 0 || F <= 0 || ldw1 < F) return LENS_ERR_SIZE;
    if ((size_t)cap_words < lens_event_capacity_words(T, F) || (cap_words & (EV_WORDS - 1))) return LENS_ERR_SIZE;
    if (N > 1024 || F > 2048) return LENS_ERR_UNSUPPORTED;
    if (ev_scale < 1 || (long long)ev_scale * F > 0x7fffffffLL) return LENS_ERR_SIZE;
    if (Q == 0) return LENS_OK;
    if (n_streams <= 0 || n_streams > Q) {
        if (carry && n_streams > Q) return LENS_ERR_SIZE;
        n_streams = carry ? 1 : Q;
    }
    FeatArgs a{d_frames, Q, HW, d_sel, N, d_codes, T, d_w1t, F, ldw1, d_vfeat, n_streams, carry ? 1 : 0,
               d_events, cap_words, d_nwords, ev_scale, d_featspk, d_active_steps};
    hipStream_t st = (hipStream_t)stream;
    const int npl = (N + 63) / 64;
    if (npl <= 1) return dispatch_feature_f<1>(a, st);
    if (npl <= 2) return dispatch_feature_f<2>(a, st);
    if (npl <= 4) return dispatch_feature_f<4>(a, st);
    return dispatch_feature_f<16>(a, st);
}

int lens_feature_currents(const uint8_t *d_frames, int Q, int HW, const int32_t *d_sel, int N,
                          const uint8_t *d_codes, int T, const float *d_w1t, int F, int ldw1,
                          float *d_currents, int ldc, void *stream)
{
    if (!d_frames || !d_sel || !d_codes || !d_w1t || !d_currents) return LENS_ERR_NULL;
    if (Q < 0 || HW <= 0 || N <= 0 || T <= 0 || F <= 0 || ldw1 < F) return LENS_ERR_SIZE;
    if (ldc < F || (ldc & 63)) return LENS_ERR_SIZE;
    if ((uintptr_t)d_currents & 15) return LENS_ERR_ALIGN;          // 16-byte stores / loads of four steps
    if (N > 1024 || F > 1024) return LENS_ERR_UNSUPPORTED;
    if (Q == 0) return LENS_OK;
    FeatArgs a{d_frames, Q, HW, d_sel, N, d_codes, T, d_w1t, F, ldw1, nullptr, Q, 0, nullptr, 0, nullptr, 1, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    const int npl = (N + 63) / 64;
    if (npl <= 1) return dispatch_currents_f<1>(a, d_currents, ldc, st);
    if (npl <= 2) return dispatch_currents_f<2>(a, d_currents, ldc, st);
    if (npl <= 4) return dispatch_currents_f<4>(a, d_currents, ldc, st);
    return dispatch_currents_f<16>(a, d_currents, ldc, st);
}

size_t lens_feature_fire_scratch_bytes(int T, int F, int n_streams)
{
    if (T <= 0 || F <= 0 || n_streams <= 0) return 0;
    return (size_t)n_streams * 2 * ((size_t)(F + 63) / 64) * (size_t)T * 64 * sizeof(float);   // two sets, used by alternate queries
}

int lens_feature_fire(const float *d_currents, int ldc, int Q, int T, int F,
                      float *d_vfeat, int n_streams,
                      uint32_t *d_events, int cap_words, uint32_t *d_nwords, int ev_scale,
                      void *d_scratch, void *stream)
{
    if (!d_currents || !d_vfeat || !d_events || !d_nwords || !d_scratch) return LENS_ERR_NULL;
    if (Q < 0 || T <= 0 || F <= 0 || ldc < F || (ldc & 63)) return LENS_ERR_SIZE;
    if ((size_t)cap_words < lens_event_capacity_words(T, F) || (cap_words & (EV_WORDS - 1))) return LENS_ERR_SIZE;
    if (ev_scale < 1 || (long long)ev_scale * F > 0x7fffffffLL) return LENS_ERR_SIZE;
    if (n_streams <= 0) n_streams = 1;
    if (n_streams > Q && Q > 0) return LENS_ERR_SIZE;
    if (((uintptr_t)d_scratch & 3) || ((uintptr_t)d_currents & 15)) return LENS_ERR_ALIGN;
    const int NW = (F + 63) / 64;
    const size_t lds = fire_lds_bytes(T, F, cap_words);
    if (NW > FIRE_MAX_WAVES || T >= (1 << 22) || lds > (size_t)LDS_BUDGET - 1024) return LENS_ERR_UNSUPPORTED;
    if (Q == 0) return LENS_OK;
    FireArgs a{d_currents, ldc, Q, T, F, d_vfeat, n_streams, d_events, cap_words, d_nwords, ev_scale,
               static_cast<float *>(d_scratch)};
    if (int e = set_lds(k_feature_fire, lds)) return e;
    hipLaunchKernelGGL(k_feature_fire, dim3(n_streams), dim3((NW + FIRE_HELPERS) * 64), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int lens_output_counts(const uint32_t *d_events, int cap_words, const uint32_t *d_nwords, int Q,
                       const float *d_w2t, int F, int P, int ldw2,
                       float *d_vout, int ldv, int n_groups, int carry,
                       float *d_counts, int ldo, void *stream)
{
    if (!d_events || !d_nwords || !d_w2t || !d_counts) return LENS_ERR_NULL;
    if (carry && !d_vout) return LENS_ERR_NULL;
    if (Q < 0 || F <= 0 || P <= 0 || ldw2 < P || ldo < P || (carry && ldv < P)) return LENS_ERR_SIZE;
    // the kernels read whole 32-word chunks (the first one before they look at the stream's length)
    if (cap_words < EV_WORDS || (cap_words & (EV_WORDS - 1))) return LENS_ERR_SIZE;
    if ((ldw2 & 255) || ((uintptr_t)d_w2t & 15)) return LENS_ERR_ALIGN;
    if (Q == 0) return LENS_OK;
    hipStream_t st = (hipStream_t)stream;
    if (carry && n_groups <= 0) n_groups = 1;
    if (carry && n_groups > Q) return LENS_ERR_SIZE;

    const OutPlan plan = output_plan(F, P, carry, n_groups);
    const int ppl = plan.ppl;
    const int pt = 64 * (ppl ? ppl : 4);
    const int ntiles = (P + pt - 1) / pt;

    // Batch mode: whole tiles for as many full rounds of the CUs as there are (a workgroup = one tile x all queries), the
    // remaining tiles split into query groups (see the kernel).  An explicit n_groups > 0 splits every tile that way.
    int threads, grid, n_whole = 0, n_main = NUM_CUS, g_main = 1;
    if (carry && plan.mode == OUT_SHARED) {
        threads = 64 * (n_groups < 16 ? n_groups : 16);
        grid = (plan.main_tiles ? plan.main_tiles : ntiles) * ((n_groups + 15) / 16);
    } else if (carry) {
        threads = plan.mode == OUT_LONE_LDS && ppl ? 256 : 64;   // lone wavefronts on LDS tiles: three more to stage the tile
        grid = ntiles * n_groups;
    } else {
        threads = 1024;
        if (n_groups <= 0) {
            // tuning override: LENS_AMD_OUTPUT_WHOLE=0 switches the whole-tile rounds off (read once)
            static const int whole = [] { const char *e = getenv("LENS_AMD_OUTPUT_WHOLE"); return e ? atoi(e) : 1; }();
            // main part: query groups per tile = the smallest power of two that makes the tiles fill the CUs, as long as a
            // workgroup keeps >= 16 queries per wavefront
            while (ntiles * g_main < NUM_CUS) g_main *= 2;
            if (whole && Q / g_main >= 16 * 16) {
                n_main = NUM_CUS / g_main;
                n_whole = (ntiles / n_main) * NUM_CUS;
            } else {
                g_main = 1;
            }
            const int rem = ntiles - (n_whole / NUM_CUS) * n_main;
            if (rem > 0) {
                // about 16 queries per wavefront (the tile is staged once per workgroup), but at least one workgroup per
                // CU; measured on Q=4096: P=1000 -> 32 groups (256 workgroups)
                const int fill = rem < NUM_CUS ? NUM_CUS / rem : 1;   // rounded down: one workgroup over 256 costs a second round
                const int load = (Q + 255) / 256;
                const int most = (Q + 15) / 16;
                n_groups = fill > load ? fill : load;
                if (n_groups > most) n_groups = most;
                if (n_groups < 1) n_groups = 1;
                if ((long long)rem * n_groups > NUM_CUS) {
                    // Several rounds of workgroups on the 256 CUs: the last round should be a full one.  Time ~ rounds / groups
                    // (a group's workgroup takes 1 / groups of the queries); 12 to 32 groups cost about the same otherwise
                    // (measured at 100 000 places: 16-24 flat, 12 +1 %, 32 +2 %).  12 501 places (an eighth of the database)
                    // go from 4.1 rounds of 16 groups to 6.96 rounds of 27.
                    double best = 0.0;
                    int best_g = n_groups;
                    const int g_lo = 12 < most ? 12 : most, g_hi = 32 < most ? 32 : most;
                    for (int g = g_lo; g <= g_hi; ++g) {
                        const long long rounds = ((long long)rem * g + NUM_CUS - 1) / NUM_CUS;
                        const double pen = 1.0 + 0.0012 * (g > 24 ? g - 24 : 0) + 0.004 * (g < 16 ? 16 - g : 0);
                        const double cost = (double)rounds / g * pen;
                        if (g == g_lo || cost < best) { best = cost; best_g = g; }
                    }
                    n_groups = best_g;
                }
            } else {
                n_groups = 1;
            }
        }
        if (n_groups > Q) n_groups = Q;
        grid = n_whole + (ntiles - (n_whole / NUM_CUS) * n_main) * n_groups;
    }
    OutArgs a{d_events, cap_words, d_nwords, Q, d_w2t, F, P, ldw2, d_vout, ldv, n_groups, carry ? (plan.mode == OUT_SHARED ? 2 : 1) : 0,
              d_counts, ldo, ntiles, n_whole, n_main, g_main, 4 * ldw2};   // row_mul: event scale 1 for the through-L2 tiles of wide layers
    if (carry && plan.mode == OUT_SHARED && plan.main_tiles) {
        // whole rounds of the CUs on the wide shared tiles, then the remaining places (columns p_main ... P - 1 of W2^T, of
        // the membranes and of the counts) as lone wavefronts on 128-place tiles: same streams, same event scale
        const int p_main = plan.main_tiles * pt;
        OutArgs m = a, t = a;
        m.P = p_main;
        m.ntiles = plan.main_tiles;
        const int e = ppl == 3 ? launch_output<3, true>(m, threads, grid, st) : launch_output<2, true>(m, threads, grid, st);
        if (e) return e;
        t.w2t += p_main;
        t.vout += p_main;
        t.counts += p_main;
        t.P = P - p_main;
        t.ntiles = (t.P + 127) / 128;
        t.carry = 1;
        return launch_output<2, true, true>(t, 256, t.ntiles * n_groups, st);
    }
    if (carry && plan.mode == OUT_LONE_L2) {
        // every tile's wavefront resident at once (see output_plan); the streams carry byte offsets of rows at the minimal ldw2
        if (ldw2 != (P + 255) / 256 * 256) return LENS_ERR_SIZE;
        hipLaunchKernelGGL(k_output_carry_l2, dim3((unsigned)grid), dim3(64), 0, st, a);
        return (int)hipGetLastError();
    }
    if (carry && plan.mode == OUT_LONE_LDS && ppl == 1) return launch_output<1, true, true>(a, threads, grid, st);
    switch (ppl) {
    case 4: return launch_output<4, true>(a, threads, grid, st);
    case 3: return launch_output<3, true>(a, threads, grid, st);
    case 2: return launch_output<2, true>(a, threads, grid, st);
    case 1: return launch_output<1, true>(a, threads, grid, st);
    default: return launch_output<4, false>(a, threads, grid, st);
    }
}

int lens_forward_counts_static(const uint8_t *d_frames, int Q, int HW, const int32_t *d_sel, int N,
                               const uint8_t *d_codes, int T, const float *d_w1t, int F, int ldw1,
                               const float *d_w2t, int P, int ldw2,
                               float *d_vfeat, float *d_vout, int ldv, int n_streams, int carry,
                               float *d_counts, int ldo, void *stream)
{
    if (Q < 0 || T <= 0 || F <= 0) return LENS_ERR_SIZE;
    // workspace layout inside the scratch array: nwords [Q] (rounded up to 256 B), then events [Q][cap]
    const size_t cap = lens_event_capacity_words(T, F);
    const size_t nwords_bytes = (((size_t)Q * 4) + 255) & ~(size_t)255;
    if (cap > 0x7fffffffu || nwords_bytes + (size_t)Q * cap * 4 > STATIC_SCRATCH_BYTES) return LENS_ERR_SCRATCH;
    unsigned char *base = nullptr;
    if (int e = static_scratch(&base)) return e;         // first call on a device resolves the array's address
    if (Q == 0) return LENS_OK;
    uint32_t *nwords = reinterpret_cast<uint32_t *>(base);
    uint32_t *events = reinterpret_cast<uint32_t *>(base + nwords_bytes);
    // every word the output kernel reads is written by the feature kernel of this call (nwords[q] and the first
    // nwords[q] words of each stream), so nothing in the array has to start at zero
    int split = carry ? n_streams : 0;
    if (carry && (split <= 0 || split > Q)) {
        if (split > Q) return LENS_ERR_SIZE;
        split = 1;
    }
    // everything lens_output_counts would refuse is refused here, before the feature layer is enqueued: a call that
    // fails must not leave the two layers' carried states out of step
    if (!d_w2t || !d_counts || (carry && !d_vout)) return LENS_ERR_NULL;
    if (P <= 0 || ldw2 < P || ldo < P || (carry && ldv < P)) return LENS_ERR_SIZE;
    if ((ldw2 & 255) || ((uintptr_t)d_w2t & 15)) return LENS_ERR_ALIGN;
    if (carry && output_plan(F, P, carry, split).mode == OUT_LONE_L2 && ldw2 != (P + 255) / 256 * 256) return LENS_ERR_SIZE;
    if (int e = lens_feature_events(d_frames, Q, HW, d_sel, N, d_codes, T, d_w1t, F, ldw1, d_vfeat, split, carry,
                                    events, (int)cap, nwords, lens_event_scale(F, P, carry, split), nullptr, nullptr, stream))
        return e;
    return lens_output_counts(events, (int)cap, nwords, Q, d_w2t, F, P, ldw2, d_vout, ldv, split, carry, d_counts, ldo,
                              stream);
}

int lens_seqmatch(const float *d_counts, int Q, int P, int ldo, int L, float *d_S, int lds, void *stream)
{
    if (!d_counts || !d_S) return LENS_ERR_NULL;
    if (L < 1 || Q < L || P < L || ldo < P || lds < P - L + 1) return LENS_ERR_SIZE;
    const int Ps = P - L + 1, Qs = Q - L + 1;
    int bx = (Ps + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_seqmatch, dim3(bx, Qs), dim3(256), 0, (hipStream_t)stream, d_counts, Ps, ldo, L, d_S, lds);
    return (int)hipGetLastError();
}

int lens_seqmatch_topk(const float *d_counts, int Q, int ldo, int n_places, int L, int K, int place_offset,
                       int32_t *d_topk_idx, float *d_topk_val, void *stream)
{
    if (!d_counts || !d_topk_idx || !d_topk_val) return LENS_ERR_NULL;
    if (L < 1 || Q < L || n_places < 1 || n_places + L - 1 > ldo || K < 1 || K > 64) return LENS_ERR_SIZE;
    const int nq = Q - L + 1;
    const dim3 grid((nq + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
#define LENS_TOPK_CASE(LT) hipLaunchKernelGGL(k_seqmatch_topk<LT>, grid, block, 0, st, d_counts, ldo, n_places, L, K, \
                                              place_offset, nq, d_topk_idx, d_topk_val)
    switch (L) {
    case 1: LENS_TOPK_CASE(1); break;
    case 2: LENS_TOPK_CASE(2); break;
    case 3: LENS_TOPK_CASE(3); break;
    case 4: LENS_TOPK_CASE(4); break;
    default: LENS_TOPK_CASE(0); break;
    }
#undef LENS_TOPK_CASE
    return (int)hipGetLastError();
}

int lens_topk_merge(const int32_t *d_idx_in, const float *d_val_in, int R, int Qs, int K,
                    int32_t *d_idx_out, float *d_val_out, void *stream)
{
    if (!d_idx_in || !d_val_in || !d_idx_out || !d_val_out) return LENS_ERR_NULL;
    if (R < 1 || Qs < 0 || K < 1 || K > 64) return LENS_ERR_SIZE;
    if (Qs == 0) return LENS_OK;
    hipLaunchKernelGGL(k_topk_merge, dim3((Qs + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_idx_in, d_val_in, R,
                       Qs, K, d_idx_out, d_val_out);
    return (int)hipGetLastError();
}


int lens_train_in_pad(int n_in)
{
    for (int p : {64, 128, 200, 208, 256, 512, 1024, 2048})   // 200: the default LENS feature count (2 x 10 x 10), no padding
        if (n_in <= p) return p;
    return 0;
}

int lens_layer_forward_clamped(const float *d_x, int S, int ldx, const float *d_w, int ldw, const float *d_thr,
                               int n_in, int n_out, float *d_y, int ldy, void *stream)
{
    if (!d_x || !d_w || !d_thr || !d_y) return LENS_ERR_NULL;
    if (S < 0 || n_in <= 0 || n_out <= 0 || ldx < n_in || ldw < n_in || ldy < n_out) return LENS_ERR_SIZE;
    if (S == 0) return LENS_OK;
    for (int s0 = 0; s0 < S; s0 += 4 * 65535) {   // grid.y is limited to 65535 blocks of 4 samples
        const int ns = (S - s0) < 4 * 65535 ? (S - s0) : 4 * 65535;
        hipLaunchKernelGGL(k_layer_forward_clamped, dim3((unsigned)((n_out + 63) / 64), (unsigned)((ns + 3) / 4)), dim3(256), 0,
                           (hipStream_t)stream, d_x + (size_t)s0 * ldx, ns, ldx, d_w, ldw, d_thr, n_in, n_out,
                           d_y + (size_t)s0 * ldy, ldy);
    }
    return (int)hipGetLastError();
}

// Lanes per weight row for a padded fan-in and a number of rows.  A lane keeps pad / lanes weights in registers; fewer
// lanes per row mean fewer wavefronts and no idle lanes in the forward chain, but more registers per lane, i.e. fewer
// resident wavefronts.  Every wavefront runs all T steps, so wavefronts beyond the resident ones run as a second round
// after the first has finished.  Model (fits the measured 100 000 x 200 layer within 5 % for 1, 2 and 4 lanes): a
// wavefront executes I = pad + 10 * weights-per-lane + 150 instructions per step; a round with n wavefronts per SIMD
// takes I * max(6, 2.2 n) cycles per step (one wavefront issues an instruction every ~6 cycles, a SIMD one every ~2.2).
static int train_lanes_per_row(int pad, int n_out)
{
    static const int forced = [] { const char *e = getenv("LENS_AMD_TRAIN_LPR"); return e ? atoi(e) : 0; }();
    int lo, hi;        // the instantiated range for this pad
    switch (pad) {
    case 64: lo = 1; hi = 1; break;
    case 128: lo = 1; hi = 2; break;
    case 200: lo = 1; hi = 2; break;
    case 208: lo = 1; hi = 4; break;
    case 256: lo = 2; hi = 4; break;
    case 512: lo = 4; hi = 8; break;
    case 1024: lo = 8; hi = 8; break;
    default: lo = 16; hi = 16; break;
    }
    if (forced >= lo && forced <= hi && (forced & (forced - 1)) == 0) return forced;
    int best = lo;
    double best_cost = 0.0;
    for (int l = lo; l <= hi; l *= 2) {
        const int hp = pad / l;
        const int occ = hp <= 64 ? 5 : hp <= 128 ? 3 : 2;   // as set by LENS_TRAIN_KERNEL
        const double instr = pad + 10.0 * hp + 150.0;
        long long waves = ((long long)n_out * l + 63) / 64;
        double cost = 0.0;
        while (waves > 0) {
            const long long now = waves < 1024LL * occ ? waves : 1024LL * occ;
            const double per_simd = (double)now / 1024.0;
            cost += instr * (per_simd * 2.2 > 6.0 ? per_simd * 2.2 : 6.0);
            waves -= now;
        }
        if (l == lo || cost < best_cost) { best = l; best_cost = cost; }
    }
    return best;
}

int lens_train_layer(float *d_w, int ldw, float *d_thr, const uint8_t *d_exc, const uint8_t *d_inh, int ldm,
                     const float *d_fire_rate, int have_rate, const float *d_x_table, const float *d_mpre_table, int ldx,
                     const int32_t *d_order, const int32_t *d_force_idx, const float *d_eta_ip, const float *d_eta_stdp,
                     int n_out, int n_in, int T, void *stream)
{
    if (!d_w || !d_thr || !d_exc || !d_inh || !d_fire_rate || !d_x_table || !d_order || !d_eta_ip || !d_eta_stdp)
        return LENS_ERR_NULL;
    const bool spk = d_force_idx != nullptr;
    if (spk && !d_mpre_table) return LENS_ERR_NULL;
    if (n_out <= 0 || n_in <= 0 || T < 0) return LENS_ERR_SIZE;
    const int pad = lens_train_in_pad(n_in);
    if (!pad) return LENS_ERR_UNSUPPORTED;
    if (ldx < pad || ldw < pad || ldm < pad) return LENS_ERR_SIZE;
    if ((ldw & 3) || ((uintptr_t)d_w & 15) || (ldx & 3) || ((uintptr_t)d_x_table & 15)) return LENS_ERR_ALIGN;
    if (T == 0) return LENS_OK;
    TrainArgs a{d_w, ldw, d_thr, d_exc, d_inh, ldm, d_fire_rate, have_rate ? 1 : 0, d_x_table, d_mpre_table, ldx,
                d_order, d_force_idx, d_eta_ip, d_eta_stdp, n_out, n_in, T};
    const dim3 block(64);
    hipStream_t st = (hipStream_t)stream;
#define LENS_TRAIN_CASE(P, LPR)                                                                          \
    if (pad == P && lpr == LPR) {                                                                        \
        const dim3 grid((n_out + 64 / LPR - 1) / (64 / LPR));                                            \
        if (spk) hipLaunchKernelGGL((k_train_layer_##P##_##LPR<true>), grid, block, 0, st, a);           \
        else hipLaunchKernelGGL((k_train_layer_##P##_##LPR<false>), grid, block, 0, st, a);              \
        return (int)hipGetLastError();                                                                   \
    }
    const int lpr = train_lanes_per_row(pad, n_out);
    LENS_TRAIN_CASE(64, 1) LENS_TRAIN_CASE(128, 1) LENS_TRAIN_CASE(128, 2) LENS_TRAIN_CASE(200, 1) LENS_TRAIN_CASE(200, 2)
    LENS_TRAIN_CASE(208, 1) LENS_TRAIN_CASE(208, 2)
    LENS_TRAIN_CASE(208, 4) LENS_TRAIN_CASE(256, 2) LENS_TRAIN_CASE(256, 4) LENS_TRAIN_CASE(512, 4) LENS_TRAIN_CASE(512, 8)
    LENS_TRAIN_CASE(1024, 8) LENS_TRAIN_CASE(2048, 16)
#undef LENS_TRAIN_CASE
    return LENS_ERR_UNSUPPORTED;
}


int lens_sad_matrix(const uint8_t *d_query, int Q, const uint8_t *d_reference, int P, int HW, float *d_dist, int ldo,
                    void *stream)
{
    if (!d_query || !d_reference || !d_dist) return LENS_ERR_NULL;
    if (Q < 0 || P < 0 || HW <= 0 || ldo < P) return LENS_ERR_SIZE;
    if ((HW & 15) || ((uintptr_t)d_query & 15) || ((uintptr_t)d_reference & 15)) return LENS_ERR_ALIGN;
    const size_t lds = (size_t)SAD_QT * HW;
    if (lds > (size_t)LDS_BUDGET) return LENS_ERR_UNSUPPORTED;
    if (Q == 0 || P == 0) return LENS_OK;
    if (int e = set_lds(k_sad_matrix, lds)) return e;
    hipLaunchKernelGGL(k_sad_matrix, dim3((P + 255) / 256, (Q + SAD_QT - 1) / SAD_QT), dim3(256), lds, (hipStream_t)stream,
                       d_query, Q, d_reference, P, HW, d_dist, ldo);
    return (int)hipGetLastError();
}


int lens_output_counts_dense(const float *d_featspk, int Q, int T, int F, int ld_f, const float *d_w2t, int P, int ldw2,
                             float *d_counts, int ldo, void *stream)
{
    if (!d_featspk || !d_w2t || !d_counts) return LENS_ERR_NULL;
    if (Q < 0 || T <= 0 || F <= 0 || ld_f < F || P <= 0 || ldw2 < P || ldo < P) return LENS_ERR_SIZE;
    if (ldw2 & 255) return LENS_ERR_ALIGN;
    if (F > 256) return LENS_ERR_UNSUPPORTED;
    if (Q == 0) return LENS_OK;
    const dim3 grid((P + 127) / 128, (Q + 31) / 32), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (F <= 64) hipLaunchKernelGGL(k_output_counts_mfma<32>, grid, block, 0, st, d_featspk, Q, T, F, ld_f, d_w2t, ldw2, P, d_counts, ldo);
    else if (F <= 128) hipLaunchKernelGGL(k_output_counts_mfma<64>, grid, block, 0, st, d_featspk, Q, T, F, ld_f, d_w2t, ldw2, P, d_counts, ldo);
    else if (F <= 200) hipLaunchKernelGGL(k_output_counts_mfma<100>, grid, block, 0, st, d_featspk, Q, T, F, ld_f, d_w2t, ldw2, P, d_counts, ldo);
    else hipLaunchKernelGGL(k_output_counts_mfma<128>, grid, block, 0, st, d_featspk, Q, T, F, ld_f, d_w2t, ldw2, P, d_counts, ldo);
    return (int)hipGetLastError();
}

}   // extern "C"

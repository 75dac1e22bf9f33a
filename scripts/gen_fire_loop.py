#!/usr/bin/env python3
"""Generates lens_amd/csrc/fire_loop.inc: gfx950 assembly of k_feature_fire's walk over the T time steps of one query
for one wavefront (64 feature neurons, one per lane, membrane in one register).  The generated file is committed;
rerun this script after editing.

Why assembly: a carried stream is ONE workgroup with one wavefront per SIMD, so the walk runs at the latency of its
instruction chain, and a branch costs a lone wavefront ~20 cycles.  As compiled from C++ a quiet step takes five
branches; here a quiet step is one taken branch, a step with current and no spike two branches that fall through.

The currents are laid out [t / 4][feature][4]: one 16-byte load per lane fetches a feature's currents of four steps.
Per step d of DEPTH unrolled steps (currents I[d] prefetched DEPTH steps ahead into a ring of registers, four per load):
    wait for I[d] (at the first step of its four)
    (QUIET_TEST only) no lane has current and no membrane sits at exactly 1.0  ->  skip (the identity on clip outputs)
    vm = v + I[d];  h = clamp(fma(vm, 0.5, 0.5));  some lane >= 1 - 2^-24 ?  -> FIRE (out of line)
    v = fma(h, 2, -1)                                                            (the clip, iaf_clip)
    (after the fourth step of a load) prefetch the four currents DEPTH steps ahead
FIRE: s = floor(max(vm, 0)); vm -= s; v = clip(vm); hot = lanes with v == 1.0; the step's spike mask (64 bits) goes to
the wavefront's row of the LDS table, one 8-byte entry per step (pre-zeroed).  Counts above 1 (rare) also store the
64 counts of the step to the wavefront's scratch row and set the step's flag word; that path drains the prefetches.
Same f32 operations in the same order as iaf_update_fast() in lens_kernels.hip: bit-identical to the one-pass kernel.

Operands: [v] membrane (in/out); [hot] SGPR pair, lanes at exactly 1.0 (in: -1 = step 0 always runs; out);
[ptr] SGPR pair: address of the query's currents; [off] 16 x the lane's feature; [lane4] 4 x lane; [ldc4] bytes of a
group of four steps (16 x ldc); [tab] LDS byte address of the wavefront's mask entry of step 0 (in/out, advanced); [flg] LDS byte address
of its flag word of step 0 (in/out, advanced); [sc] SGPR pair: the wavefront's scratch rows [T][64] floats; [T] steps.
Up to DEPTH steps behind the query's last are read (and ignored): the buffer carries DEPTH rows of slack.
Clobbers: v[64:64+DEPTH-1] (the ring), v[96:99], s[36:51], vcc, scc.
"""
import os

DEPTH = 32
# QUIET_TEST: skip the update of a step in which no lane has current and no membrane sits at exactly 1.0 (the identity
# on clip outputs).  Off: every step runs the literal update -- one dependent compare-and-branch less per step, which a
# lone wavefront pays for with latency: 14.7 -> 13.2 us per query at the benchmark's density (31 % of its steps are quiet;
# on the bundled frames hardly any).
QUIET_TEST = os.environ.get("LENS_GEN_FIRE_QUIET", "0") == "1"
GROUP = 4                      # steps whose currents sit side by side in memory: one 16-byte load per lane
RING = 64                      # first VGPR of the ring
VM, H, A0, A1 = 96, 97, 98, 99
S_N, S_T, S_G = 36, 37, 38     # steps left in the tail group, first step of the current group, full groups left
S_TMP = 40                     # pair
S_M2 = 42                      # pair: lanes with a count above 1
S_ROW = 44                     # pair: scratch row address (rare path)
S_X = 46
S_PTR, S_SC = 48, 50           # pairs: next row to prefetch; the wavefront's scratch rows


def gen():
    hot_lines, cold = [], []
    e = hot_lines.append

    def group(name, checked):
        """DEPTH steps; checked: stop after s[S_N] steps (the last, partial group)"""
        for d in range(DEPTH):
            L = lambda n: f"{n}{name}{d}_%="
            I = f"v{RING + d}"
            if checked:
                e(f"s_cmp_le_u32 s{S_N}, {d}")
                e(f"s_cbranch_scc1 DONE_%=")
            if d % GROUP == 0:
                e(f"s_waitcnt vmcnt({DEPTH // GROUP - 1})")               # the 16-byte load of this group of four steps
            if QUIET_TEST:
                e(f"v_cmp_neq_f32_e32 vcc, 0, {I}")
                e(f"s_or_b64 s[{S_TMP}:{S_TMP + 1}], vcc, %[hot]")          # scc = some current, or some membrane at 1.0
                e(f"s_cbranch_scc0 {L('NEXT')}")
            e(f"v_add_f32_e32 v{VM}, %[v], {I}")
            e(f"v_cmp_le_f32_e32 vcc, 0x3f7fffff, v{VM}")        # 1 - 2^-24 included: the clip rounds it up to 1.0
            e(f"v_fma_f32 v{H}, v{VM}, 0.5, 0.5 clamp")
            e(f"s_cbranch_vccnz {L('FIRE')}")
            e(f"v_fma_f32 %[v], v{H}, 2.0, -1.0")
            if QUIET_TEST:
                e(f"s_mov_b64 %[hot], 0")
            e(f"{L('NEXT')}:")
            if d % GROUP == GROUP - 1:                                  # steps t + DEPTH - 3 .. t + DEPTH into the four ring registers just consumed
                g0 = RING + d - (GROUP - 1)
                e(f"global_load_dwordx4 v[{g0}:{g0 + GROUP - 1}], %[off], s[{S_PTR}:{S_PTR + 1}]")
                e(f"s_add_u32 s{S_PTR}, s{S_PTR}, %[ldc4]")
                e(f"s_addc_u32 s{S_PTR + 1}, s{S_PTR + 1}, 0")

            c = cold.append
            c(f"{L('FIRE')}:")
            c(f"v_max_f32_e32 v{H}, 0, v{VM}")
            c(f"v_floor_f32_e32 v{H}, v{H}")                              # spike count of the lane
            c(f"v_sub_f32_e32 v{VM}, v{VM}, v{H}")
            c(f"v_cmp_lt_f32_e32 vcc, 0, v{H}")                           # spike mask
            c(f"v_fma_f32 v{VM}, v{VM}, 0.5, 0.5 clamp")
            c(f"v_cmp_lt_f32_e64 s[{S_M2}:{S_M2 + 1}], 1.0, v{H}")
            c(f"v_fma_f32 %[v], v{VM}, 2.0, -1.0")
            c(f"v_mov_b32_e32 v{A0}, vcc_lo")
            c(f"v_mov_b32_e32 v{A1}, vcc_hi")
            if QUIET_TEST:
                c(f"v_cmp_eq_f32_e64 %[hot], 1.0, %[v]")
            # (the 1 - 2^-24 corner fires nothing: it writes the zero mask the entry already holds)
            c(f"ds_write_b64 %[tab], v[{A0}:{A1}] offset:{8 * d}")        # every lane writes the same entry
            c(f"s_cmp_eq_u64 s[{S_M2}:{S_M2 + 1}], 0")
            c(f"s_cbranch_scc1 {L('NEXT')}")
            # rare: counts above 1 -> the step's 64 counts to the scratch row, flag word 1; drains the prefetch queue
            c(f"s_add_u32 s{S_X}, s{S_T}, {d}")
            c(f"s_lshl_b32 s{S_X}, s{S_X}, 8")
            c(f"s_add_u32 s{S_ROW}, s{S_SC}, s{S_X}")
            c(f"s_addc_u32 s{S_ROW + 1}, s{S_SC + 1}, 0")
            c(f"global_store_dword %[lane4], v{H}, s[{S_ROW}:{S_ROW + 1}]")
            c(f"v_mov_b32_e32 v{A1}, 1")
            c(f"ds_write_b32 %[flg], v{A1} offset:{4 * d}")
            c(f"s_waitcnt vmcnt(0)")
            c(f"s_branch {L('NEXT')}")

    e(f"s_mov_b64 s[{S_PTR}:{S_PTR + 1}], %[ptr]")
    e(f"s_mov_b64 s[{S_SC}:{S_SC + 1}], %[sc]")
    # ring fill: steps 0 .. DEPTH-1 (steps beyond T-1 exist as slack)
    for d in range(0, DEPTH, GROUP):
        e(f"global_load_dwordx4 v[{RING + d}:{RING + d + GROUP - 1}], %[off], s[{S_PTR}:{S_PTR + 1}]")
        e(f"s_add_u32 s{S_PTR}, s{S_PTR}, %[ldc4]")
        e(f"s_addc_u32 s{S_PTR + 1}, s{S_PTR + 1}, 0")
    e(f"s_mov_b32 s{S_T}, 0")
    e(f"s_lshr_b32 s{S_G}, %[T], {DEPTH.bit_length() - 1}")              # full groups
    e(f"s_and_b32 s{S_N}, %[T], {DEPTH - 1}")                            # steps of the last, partial group
    e(f"s_cmp_eq_u32 s{S_G}, 0")
    e(f"s_cbranch_scc1 TAIL_%=")
    e("LOOP_%=:")
    group("M", False)
    e(f"s_add_u32 s{S_T}, s{S_T}, {DEPTH}")
    e(f"v_add_u32_e32 %[tab], {8 * DEPTH}, %[tab]")
    e(f"v_add_u32_e32 %[flg], {4 * DEPTH}, %[flg]")
    e(f"s_sub_u32 s{S_G}, s{S_G}, 1")
    e(f"s_cmp_lg_u32 s{S_G}, 0")
    e(f"s_cbranch_scc1 LOOP_%=")
    e("TAIL_%=:")
    group("X", True)
    e(f"s_branch DONE_%=")
    return hot_lines + cold + ["DONE_%=:", "s_waitcnt vmcnt(0) lgkmcnt(0)"]


def main():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "lens_amd", "csrc")
    path = os.path.join(out_dir, "fire_loop.inc")
    lines = gen()
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_fire_loop.py -- do not edit.  gfx950 step walk of k_feature_fire.\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    print(path, len(lines), "lines")


if __name__ == "__main__":
    main()

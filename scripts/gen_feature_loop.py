#!/usr/bin/env python3
"""Generates lens_amd/csrc/feature_loop_n2f4.inc: gfx950 assembly of k_feature_events' time-step loop for one
query (instantiation NPL=2, FPL=4, tables in LDS: up to 128 inputs, up to 256 features).  The generated file is
committed; rerun this script after editing.

Why assembly: the loop is issue-bound (4 waves/SIMD next to the 105-134 KB of tables) and the compiler spends
~40 scalar instructions per step on control-flow bookkeeping.  Hand-scheduled, a step is:
  * thresholds of the next step prefetched from LDS; `px >= code` -> two 64-bit input masks;
  * a step without input spikes and without a membrane at exactly 1.0 is skipped outright (every membrane is a
    clip output by then, for which v += 0; fire; clip is the identity);
  * active inputs in ascending n: `acc[f] += W1T[n][f]` (== fmaf(1, w, acc)), two ds_read2st64_b32 per input;
  * vm = v + acc; if no lane reaches 1 - 2^-24: clip only (two FMAs per register) and the step is empty;
    otherwise the literal update s = floor(max(vm, 0)); vm -= s; v = clip(vm) and the compacted event writer;
  * the spike-stream rules of lens_kernels.hip (LAST sign, EMPTY_NEXT, explicit empty step, CHUNK_MIXED).
The arithmetic is the same sequence of f32 operations as the C++ path (iaf_update_fast), so results are bit-identical.

Operands: [v0..v3] membranes (in/out), [count] events written (in/out, starts at 0), [px0],[px1] pixel values,
[caddr] LDS byte address of this lane's threshold code at step 0 (in/out), [wa0],[wa1] LDS byte address of
W1T[0][lane] / W1T[64][lane], [fo0..fo3] offset words of this lane's four features, [ev] stream pointer, [T] steps.
Clobbers: s[36:67], v[40:67], vcc, scc.  EXEC is narrowed for the stores and restored to all ones.
"""
import os

S_T, S_FLAG, S_LAST, S_BASE, S_LASTPOS, S_TMP, S_TMP2, S_N, S_TOTAL = 36, 37, 38, 39, 40, 41, 42, 43, 44
M_IN = (46, 48)                     # input masks (SGPR pairs)
M_SP = (50, 52, 54, 56)             # spike masks per feature register
S_HOT = 58                          # pair: lanes whose membrane is exactly 1.0
S_CNT = (60, 61, 62, 63)            # spikes per feature register
S_EV = 64                           # pair: stream pointer
S_PAIR = 66                         # pair: 64-bit scratch
C0, C1, C0N, C1N = 40, 41, 42, 43   # threshold codes, this step / next step
ACC = (44, 45, 46, 47)              # input current, then vm
W = (48, 49, 50, 51)
W2 = (64, 65, 66, 67)
ADDR = 52
SP = (53, 54, 55, 56)
TMP, POS = 57, 58
D0, D1 = 60, 61                     # event words for the store (consecutive pair)
AUX, AUX2 = 62, 63


def gen():
    out = []
    e = out.append
    sfx = [""]
    ool = {}                                    # out-of-line blocks of a step copy, emitted behind the loop

    def L(n): return f"{n}{sfx[0]}_%="          # label local to one copy of the step body
    def G(n): return f"{n}_%="                  # shared label
    def pair(s): return f"s[{s}:{s + 1}]"
    V = ["%[v0]", "%[v1]", "%[v2]", "%[v3]"]
    FO = ["%[fo0]", "%[fo1]", "%[fo2]", "%[fo3]"]

    def lane0_store_dword(byte_off_sgpr, value_sgpr):
        e(f"v_mov_b32 v{AUX}, s{byte_off_sgpr}")
        e(f"v_mov_b32 v{AUX2}, s{value_sgpr}")
        e("s_mov_b64 exec, 1")
        e(f"global_store_dword v{AUX}, v{AUX2}, {pair(S_EV)}")
        e("s_mov_b64 exec, -1")

    def issue_input(i, regs):
        """LDS reads of the next active input of group i into `regs` (and clear its mask bit)"""
        e(f"s_ff1_i32_b64 s{S_N}, {pair(M_IN[i])}")
        e(f"v_lshl_add_u32 v{ADDR}, s{S_N}, 10, %[wa{i}]")      # row n of W1T: 256 floats
        e(f"ds_read2st64_b32 v[{regs[0]}:{regs[1]}], v{ADDR} offset1:1")
        e(f"ds_read2st64_b32 v[{regs[2]}:{regs[3]}], v{ADDR} offset0:2 offset1:3")
        e(f"s_bitset0_b64 {pair(M_IN[i])}, s{S_N}")

    def add_input(regs, outstanding):
        """acc += regs, in order; `outstanding` = LDS reads issued after these"""
        e(f"s_waitcnt lgkmcnt({outstanding})")
        e(f"v_add_f32 v{ACC[0]}, v{ACC[0]}, v{regs[0]}")
        e(f"v_add_f32 v{ACC[1]}, v{ACC[1]}, v{regs[1]}")
        e(f"v_add_f32 v{ACC[2]}, v{ACC[2]}, v{regs[2]}")
        e(f"v_add_f32 v{ACC[3]}, v{ACC[3]}, v{regs[3]}")

    def step(cur, nxt, name, after):
        """one time step; cur / nxt = (code reg of inputs 0..63, of inputs 64..127) for this / the next step.
        `after`: emitted at the end of the step (loop control)."""
        sfx[0] = name
        e(f"{L('STEP')}:")
        e(f"ds_read_u8 v{nxt[0]}, %[caddr] offset:128")     # next step's thresholds (one step past the table: unused)
        e(f"ds_read_u8 v{nxt[1]}, %[caddr] offset:192")
        e("s_waitcnt lgkmcnt(2)")
        e(f"v_cmp_ge_u32_e64 {pair(M_IN[0])}, %[px0], v{cur[0]}")
        e(f"v_cmp_ge_u32_e64 {pair(M_IN[1])}, %[px1], v{cur[1]}")
        e(f"v_add_u32 %[caddr], 0x80, %[caddr]")
        # quiet step?  no input spike, no membrane at exactly 1.0 (tested out of line)
        e(f"s_or_b64 {pair(S_PAIR)}, {pair(M_IN[0])}, {pair(M_IN[1])}")      # scc = some input spikes
        e(f"s_cbranch_scc0 {L('NOIN')}")
        # The step's first active input is read straight into the accumulator (fl(0 + w) is w; a -0 weight leaves +0
        # instead of -0 there, which v + acc cannot see: a clipped v is never -0, and a carried-in -0 with no other
        # input ends as clip(+-0) = +0 either way): no zeroing and no adds for it.  Out of line: the first input in the
        # upper group (none in the lower), and a step with no input at all (a membrane at exactly 1.0 brought it here).
        e(f"s_cmp_eq_u64 {pair(M_IN[0])}, 0")
        e(f"s_cbranch_scc1 {L('FIRST1')}")
        issue_input(0, ACC)
        o = ool.setdefault(name, []).append
        o(f"{L('FIRST1')}:")                                     # no input in the lower group, so one in the upper
        lines_before = len(out)                                 # issue_input emits into the main stream: move its lines
        issue_input(1, ACC)
        moved = out[lines_before:]
        del out[lines_before:]
        for ln in moved:
            o(ln)
        o(f"s_branch {L('GRP1')}")
        o(f"{L('NOIN')}:")                                       # no input spike: quiet unless a membrane is at 1.0
        o(f"s_cmp_eq_u64 {pair(S_HOT)}, 0")
        o(f"s_cbranch_scc1 {L('EMPTY')}")
        for j in range(4):
            o(f"v_mov_b32 v{ACC[j]}, 0")
        o(f"s_branch {L('IN_DONE1')}")
        # active inputs in ascending n; the reads of the next input are in flight while the current one is added
        for i in range(2):
            e(f"{L(f'GRP{i}')}:")
            e(f"s_cmp_eq_u64 {pair(M_IN[i])}, 0")
            e(f"s_cbranch_scc1 {L(f'IN_DONE{i}')}")
            issue_input(i, W)
            e(f"{L(f'IN{i}')}:")
            e(f"s_cmp_eq_u64 {pair(M_IN[i])}, 0")
            e(f"s_cbranch_scc1 {L(f'IN_LASTA{i}')}")
            issue_input(i, W2)
            add_input(W, 2)
            e(f"s_cmp_eq_u64 {pair(M_IN[i])}, 0")
            e(f"s_cbranch_scc1 {L(f'IN_LASTB{i}')}")
            issue_input(i, W)
            add_input(W2, 2)
            e(f"s_branch {L(f'IN{i}')}")
            e(f"{L(f'IN_LASTA{i}')}:")
            add_input(W, 0)
            e(f"s_branch {L(f'IN_DONE{i}')}")
            e(f"{L(f'IN_LASTB{i}')}:")
            add_input(W2, 0)
            e(f"{L(f'IN_DONE{i}')}:")
        # vm = v + I
        e("s_waitcnt lgkmcnt(0)")                               # a single input: its reads went to the accumulator
        for j in range(4):
            e(f"v_add_f32 v{ACC[j]}, {V[j]}, v{ACC[j]}")
        e(f"v_max3_f32 v{TMP}, v{ACC[0]}, v{ACC[1]}, v{ACC[2]}")
        e(f"v_max_f32 v{TMP}, v{TMP}, v{ACC[3]}")
        e(f"v_cmp_le_f32_e32 vcc, 0x3f7fffff, v{TMP}")      # someone at or above 1 - 2^-24 ?
        e(f"s_cbranch_vccnz {L('FIRE')}")
        # nobody fires: clip only, the step is empty, no membrane can be 1.0
        for j in range(4):
            e(f"v_fma_f32 v{ACC[j]}, v{ACC[j]}, 0.5, 0.5 clamp")
        for j in range(4):
            e(f"v_fma_f32 {V[j]}, v{ACC[j]}, 2.0, -1.0")
        e(f"s_mov_b64 {pair(S_HOT)}, 0")

        # ---- a step without feature spikes (shared by the quiet path)
        e(f"{L('EMPTY')}:")
        e(f"s_cmp_lg_u32 s{S_FLAG}, 0")
        e(f"s_cbranch_scc1 {L('NEXT')}")
        e(f"s_cmp_eq_u32 s{S_T}, 0")
        e(f"s_cbranch_scc1 {L('EXPLICIT')}")
        e(f"s_and_b32 s{S_TMP}, s{S_LAST}, 0x7fffffff")
        e(f"s_cmp_lt_u32 s{S_TMP}, 0x4b000000")
        e(f"s_cbranch_scc0 {L('EXPLICIT')}")
        # flag the previous LAST event: word 2*(count-1)+1
        e(f"s_or_b32 s{S_LAST}, s{S_LAST}, 1")
        e(f"s_lshl_b32 s{S_TMP}, %[count], 3")
        e(f"s_add_u32 s{S_TMP}, s{S_TMP}, -4")
        lane0_store_dword(S_TMP, S_LAST)
        e(f"s_mov_b32 s{S_FLAG}, 1")
        e(f"s_branch {L('NEXT')}")
        e(f"{L('EXPLICIT')}:")                                # explicit empty event {0, -0.0}, its chunk is mixed
        e(f"s_lshl_b32 s{S_TMP}, %[count], 3")
        e(f"v_mov_b32 v{AUX}, s{S_TMP}")
        e(f"v_mov_b32 v{D0}, 0")
        e(f"v_bfrev_b32 v{D1}, 1")                            # 0x80000000
        e(f"s_and_b32 s{S_TMP2}, s{S_TMP}, 0xffffff80")       # byte offset of the chunk's first event (16 events x 8 B)
        e(f"v_mov_b32 v{AUX2}, s{S_TMP2}")
        e("s_mov_b64 exec, 1")
        e(f"global_store_dwordx2 v{AUX}, v[{D0}:{D1}], {pair(S_EV)}")
        e(f"global_atomic_or v{AUX2}, v{D1}, {pair(S_EV)}")
        e("s_mov_b64 exec, -1")
        e(f"s_add_u32 %[count], %[count], 1")
        e(f"s_brev_b32 s{S_LAST}, 1")
        e(f"s_branch {L('NEXT')}")

        # ---- some lane may fire: the literal update
        e(f"{L('FIRE')}:")
        for j in range(4):
            e(f"v_max_f32 v{SP[j]}, 0, v{ACC[j]}")
        for j in range(4):
            e(f"v_floor_f32 v{SP[j]}, v{SP[j]}")
        for j in range(4):
            e(f"v_sub_f32 v{ACC[j]}, v{ACC[j]}, v{SP[j]}")
            e(f"v_cmp_lt_f32_e64 {pair(M_SP[j])}, 0, v{SP[j]}")
        for j in range(4):
            e(f"v_fma_f32 v{ACC[j]}, v{ACC[j]}, 0.5, 0.5 clamp")
        for j in range(4):
            e(f"v_fma_f32 {V[j]}, v{ACC[j]}, 2.0, -1.0")
        # hot: some membrane exactly 1.0 (v <= 1 always)
        e(f"v_max3_f32 v{TMP}, {V[0]}, {V[1]}, {V[2]}")
        e(f"v_max_f32 v{TMP}, v{TMP}, {V[3]}")
        e(f"v_cmp_le_f32_e64 {pair(S_HOT)}, 1.0, v{TMP}")
        for j in range(4):
            e(f"s_bcnt1_i32_b64 s{S_CNT[j]}, {pair(M_SP[j])}")
        e(f"s_add_u32 s{S_TOTAL}, s{S_CNT[0]}, s{S_CNT[1]}")
        e(f"s_add_u32 s{S_TOTAL}, s{S_TOTAL}, s{S_CNT[2]}")
        e(f"s_add_u32 s{S_TOTAL}, s{S_TOTAL}, s{S_CNT[3]}")
        e(f"s_cmp_eq_u32 s{S_TOTAL}, 0")
        e(f"s_cbranch_scc1 {L('EMPTY')}")                      # (only the 1 - 2^-24 corner gets here without spikes)
        # event writer
        e(f"s_mov_b32 s{S_BASE}, %[count]")
        e(f"s_add_u32 s{S_LASTPOS}, %[count], s{S_TOTAL}")
        e(f"s_add_u32 s{S_LASTPOS}, s{S_LASTPOS}, -1")
        for j in range(4):
            e(f"s_cmp_eq_u32 s{S_CNT[j]}, 0")
            e(f"s_cbranch_scc1 {L(f'WR_DONE{j}')}")
            e(f"v_mbcnt_lo_u32_b32 v{POS}, s{M_SP[j]}, 0")
            e(f"v_mbcnt_hi_u32_b32 v{POS}, s{M_SP[j] + 1}, v{POS}")
            e(f"v_add_u32 v{POS}, s{S_BASE}, v{POS}")
            e(f"v_cmp_eq_u32_e32 vcc, s{S_LASTPOS}, v{POS}")
            e(f"v_or_b32 v{D1}, 0x80000000, v{SP[j]}")
            e(f"v_mov_b32 v{D0}, {FO[j]}")
            e(f"v_lshlrev_b32 v{POS}, 3, v{POS}")                  # (also the wait states between v_cmp and v_cndmask)
            e(f"v_cndmask_b32_e32 v{D1}, v{SP[j]}, v{D1}, vcc")
            e(f"s_mov_b64 exec, {pair(M_SP[j])}")
            e(f"global_store_dwordx2 v{POS}, v[{D0}:{D1}], {pair(S_EV)}")
            e("s_mov_b64 exec, -1")
            e(f"s_add_u32 s{S_BASE}, s{S_BASE}, s{S_CNT[j]}")
            e(f"{L(f'WR_DONE{j}')}:")
        e(f"s_mov_b32 s{S_LAST}, 0xbf800000")                   # a single spike, LAST: all but <1 % of the steps
        e(f"v_or3_b32 v{TMP}, v{SP[0]}, v{SP[1]}, v{SP[2]}")
        e(f"v_or_b32 v{TMP}, v{TMP}, v{SP[3]}")
        e(f"v_cmp_lt_u32_e32 vcc, 1.0, v{TMP}")                 # OR of the count patterns above 1.0f's <=> some count > 1
        e(f"s_cbranch_vccnz {L('MULTI')}")
        e(f"{L('WROTE')}:")
        e(f"s_add_u32 %[count], %[count], s{S_TOTAL}")
        e(f"s_mov_b32 s{S_FLAG}, 0")

        # ---- next step
        e(f"{L('NEXT')}:")
        e(f"s_add_u32 s{S_T}, s{S_T}, 1")
        e(f"s_cmp_lt_u32 s{S_T}, %[T]")
        after()

    def multi(name):
        """rare: some count exceeds 1 -> mark the chunks, remember the last event's count word"""
        sfx[0] = name
        for ln in ool.get(name, []):
            e(ln)
        e(f"{L('MULTI')}:")
        e(f"s_mov_b32 s{S_BASE}, %[count]")
        e(f"v_bfrev_b32 v{D1}, 1")
        for j in range(4):
            e(f"v_cmp_lt_f32_e32 vcc, 1.0, v{SP[j]}")
            e(f"v_mbcnt_lo_u32_b32 v{POS}, s{M_SP[j]}, 0")
            e(f"v_mbcnt_hi_u32_b32 v{POS}, s{M_SP[j] + 1}, v{POS}")
            e(f"v_add_u32 v{POS}, s{S_BASE}, v{POS}")
            e(f"v_lshlrev_b32 v{POS}, 3, v{POS}")
            e(f"v_and_b32 v{POS}, 0xffffff80, v{POS}")
            e("s_mov_b64 exec, vcc")
            e(f"global_atomic_or v{POS}, v{D1}, {pair(S_EV)}")
            e("s_mov_b64 exec, -1")
            e(f"s_add_u32 s{S_BASE}, s{S_BASE}, s{S_CNT[j]}")
            e(f"s_cmp_eq_u32 s{S_CNT[j]}, 0")
            e(f"s_cbranch_scc1 {L(f'ML_DONE{j}')}")
            e(f"s_flbit_i32_b64 s{S_TMP}, {pair(M_SP[j])}")
            e(f"s_sub_u32 s{S_TMP}, 63, s{S_TMP}")
            e("s_nop 3")                                        # SGPR written by SALU used as v_readlane lane select
            e(f"v_readlane_b32 s{S_LAST}, v{SP[j]}, s{S_TMP}")
            e("s_nop 0")
            e(f"s_or_b32 s{S_LAST}, s{S_LAST}, 0x80000000")
            e(f"{L(f'ML_DONE{j}')}:")
        e(f"s_branch {L('WROTE')}")

    # ---------------- prologue
    e(f"s_mov_b64 {pair(S_EV)}, %[ev]")
    e(f"s_mov_b32 s{S_T}, 0")
    e(f"s_mov_b32 s{S_FLAG}, 0")
    e(f"s_mov_b32 s{S_LAST}, 0")
    e(f"s_mov_b64 {pair(S_HOT)}, -1")                 # step 0 always runs
    e(f"ds_read_u8 v{C0}, %[caddr]")
    e(f"ds_read_u8 v{C1}, %[caddr] offset:64")
    # the loop is unrolled by two: the threshold registers of "this" and "the next" step swap roles
    step((C0, C1), (C0N, C1N), "A", lambda: e(f"s_cbranch_scc0 {G('DONE')}"))
    step((C0N, C1N), (C0, C1), "B", lambda: (e(f"s_cbranch_scc1 STEPA_%="), e(f"s_branch {G('DONE')}")))
    multi("A")
    multi("B")
    e(f"{G('DONE')}:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return out


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "lens_amd", "csrc", "feature_loop_n2f4.inc")
    lines = gen()
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_feature_loop.py -- do not edit.  gfx950 step loop of k_feature_events<2, 4, true>.\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    print(f"wrote {path}: {len(lines)} lines")


if __name__ == "__main__":
    main()

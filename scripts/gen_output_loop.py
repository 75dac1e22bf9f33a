#!/usr/bin/env python3
"""Generates lens_amd/csrc/output_loop_p{1,2,3,4}.inc: gfx950 assembly of k_output_counts' chunk loop for one query
against an LDS weight tile of 64, 128, 192 or 256 places (one to four places per lane).  The generated files
are committed; rerun this script after editing.

Why assembly: the loop runs at the issue rate of its instruction mix (4 waves per SIMD next to the 100-150 KB
tile), and the shape that needs the fewest instructions and taken branches cannot be written in C++ -- the
compiler's control-flow structurizer turns it into flag bookkeeping.  The shape is two chains of code per chunk:
  A<i>  accumulates event i (acc += count * w) and falls through to A<i+1>; the LAST event of a time step
        leaves the chain for S<i>;
  S<i>  ends the step (v += acc; any lane at threshold -> spike path, out of line; clip), then takes event i+1
        as the first of the next step: its weights REPLACE the accumulator (no zeroing, no add: fmaf(c, w, +0)
        equals c * w up to the sign of a zero, which v + acc cannot see because a clipped v is never -0).  If
        event i+1 ends its step too, S<i+1> follows directly; else the code rejoins the first chain at A<i+2>.
Per chunk: 16 x (address, LDS read(s)), wait, scalar prefetch of the next chunk into the other SGPR buffer, then
SALU/VALU-only integration while the prefetch is in flight.  Chunks whose counts are all 1 (EV_CHUNK_MIXED
clear) use plain adds; the others multiply by the count held in an SGPR, in a variant without the per-event test for
counts of 2^23 and more when the OR of the chunk's count words rules those out.
The arithmetic is the same sequence of f32 operations as the C++ path (integrate_chunk), so results are
bit-identical.

Operands: [v0..] membranes, [c0..] spike counters (in/out), [soff] byte offset of the current chunk (in/out,
starts at 0), [nch] chunks left (out; loaded from [nwp], the address of the stream's length in words), [ev] stream pointer, [lane] LDS byte address of this lane's first
weights in row 0 (three places: of its pair of places; the third place's plane starts at LDS address 0),
[thr] 1 - 2^-24 (a VGPR: a compare with an SGPR operand measured 3 % slower);
temporaries [a0..] accumulator, [t], [t2], [sc], [so] scalars.
Clobbers: s[36:99] (the two chunk buffers), the weights' VGPRs (v64.. upwards: 16, 32 or 64 registers; three
places: v[64:95] and v[96:111], and m0), vcc, scc.
"""
import os

EA, EB = 36, 68                 # first SGPR of the two 32-word chunk buffers
WPAIR, WTHIRD = 64, 96          # weights: event i -> v[64+2i : 65+2i] and (three places) v[96+i]
EV_CHUNK = 16
READ_GROUP = 16                 # events whose addresses are computed before their reads are issued
PAIR_FIRST = True               # a step's first two events enter the accumulator with one add per place
# Timing probes only (wrong results): LENS_GEN_PROBE=nowait drops the wait for the weight reads (how much of a chunk's
# time is their exposed latency), =nolds drops the reads themselves.  Never set for a committed .inc file.
PROBE = os.environ.get("LENS_GEN_PROBE", "")
# Experiment, LENS_GEN_JUMP=1 (with -DLENS_OUTPUT_JUMP for lens_kernels.hip): chunks whose counts are all 1 take one
# computed jump per time step instead of a test per event (jump_chunk).  Bit-exact, and measured slower: 61 % fewer
# branch and 38 % more scalar instructions, +5.3 % time at the benchmark's density, +0.5 % on the bundled frames --
# an s_setpc_b64 costs about as much as the ten not-taken test-and-branch pairs it replaces.  Never set for a committed file.
JUMP = os.environ.get("LENS_GEN_JUMP", "0") == "1"
COPY_STRIDE_LOG2 = 9            # byte stride of the 17 code copies a computed jump selects from


def gen(ppl):
    hot, warm, cold = [], [], []   # the loop body; code reached once per step or mixed chunk; rare code
    V = [f"%[v{k}]" for k in range(ppl)]
    C = [f"%[c{k}]" for k in range(ppl)]
    A = [f"%[a{k}]" for k in range(ppl)]

    def W(i):
        if ppl == 1:
            return [f"v{WPAIR + i}"]
        if ppl == 4:
            return [f"v{WPAIR + 4 * i + k}" for k in range(4)]
        regs = [f"v{WPAIR + 2 * i}", f"v{WPAIR + 2 * i + 1}"]
        if ppl == 3:
            regs.append(f"v{WTHIRD + i}")
        return regs

    def read(i, ad):
        w = W(i)
        if ppl == 1:
            return [f"ds_read_b32 {w[0]}, {ad}"]
        if ppl == 4:
            return [f"ds_read_b128 v[{WPAIR + 4 * i}:{WPAIR + 4 * i + 3}], {ad}"]
        r = [f"ds_read_b64 v[{WPAIR + 2 * i}:{WPAIR + 2 * i + 1}], {ad}"]
        if ppl == 3:
            r.append(f"ds_read_addtid_b32 {w[2]}")
        return r

    def fire_max():      # [t] = largest membrane of the lane's places (one place: the compare reads it directly)
        if ppl == 1:
            return [], V[0]
        if ppl == 2:
            return [f"v_max_f32 %[t], {V[0]}, {V[1]}"], "%[t]"
        r = [f"v_max3_f32 %[t], {V[0]}, {V[1]}, {V[2]}"]
        if ppl == 4:
            r.append(f"v_max_f32 %[t], %[t], {V[3]}")
        return r, "%[t]"

    def jump_chunk(e, c, lab, W, cnt):
        """Integration of a chunk whose counts are all 1, without a test per event.  The running sum of a step lives in
        the weight registers themselves (body E<k>: w[k] += w[k-1]; a step's first event costs nothing), and the code
        exists in 17 copies at a fixed byte stride: copy j is E0 .. Ej followed by the end of a step at event j, copy 16
        is E0 .. E15 for a step that goes on in the next chunk ([a] carries its sum; E0 is w[0] += [a], with [a] = 0
        after a step that ended with the chunk).  The end of step j finds the next LAST event j' in [m] (s_ff1) and
        jumps to body E<j+2> of copy j' -- the new step starts at event j+1 -- so a step of n events costs n - 1 adds
        per place, one spike test and ONE jump.  Same f32 operations in the same order as chains A / S."""
        bs = 4 * ppl                                             # bytes of one body: ppl x v_add_f32_e32
        assert (EV_CHUNK + 1) * bs + 160 <= (1 << COPY_STRIDE_LOG2)
        copy0 = lab("COPY0")
        # dispatch of the chunk's first step: always through E0 (adds the carried sum)
        e(f"{lab('A0')}:")
        e("s_ff1_i32_b32 %[x], %[m]")
        e(f"s_lshl_b32 %[x], %[x], {COPY_STRIDE_LOG2}")
        e(f"s_add_u32 %[x], %[x], {copy0}-ANCHOR_%=")
        e("s_add_u32 s100, %[x], %[bl]")
        e("s_addc_u32 s101, %[bh], 0")
        e("s_setpc_b64 s[100:101]")
        for j in range(EV_CHUNK + 1):
            e(f".p2align {COPY_STRIDE_LOG2}")
            if j == 0:
                e(f"{copy0}:")
            last = min(j, EV_CHUNK - 1)
            for k in range(last + 1):
                w = W(k)
                for q in range(ppl):
                    if k == 0:
                        e(f"v_add_f32_e32 {w[q]}, {A[q]}, {w[q]}")
                    elif j == EV_CHUNK and k == EV_CHUNK - 1:      # the open step's sum leaves the chunk in [a]
                        e(f"v_add_f32_e32 {A[q]}, {W(k - 1)[q]}, {w[q]}")
                    else:
                        e(f"v_add_f32_e32 {w[q]}, {W(k - 1)[q]}, {w[q]}")
            if j == EV_CHUNK:
                break                                            # falls through into the chunk's tail
            # ---- end of the step whose last event is j ----
            w = W(j)
            if j == EV_CHUNK - 1:
                e(f"{lab('SE15')}:")
            sal = []                                             # the jump's scalar chain, interleaved with the VALU work
            if j < EV_CHUNK - 2:
                sal = [f"s_and_b32 %[x], %[m], {hex(0x1ffff & ~((2 << j) - 1))}",
                       "s_ff1_i32_b32 %[x], %[x]",
                       f"s_lshl_b32 %[x], %[x], {COPY_STRIDE_LOG2}",
                       f"s_add_u32 %[x], %[x], {copy0}-ANCHOR_%=+{(j + 2) * bs}",
                       "s_add_u32 s100, %[x], %[bl]",
                       "s_addc_u32 s101, %[bh], 0"]
            mx, top = fire_max()
            val = [f"v_add_f32_e32 {V[q]}, {V[q]}, {w[q]}" for q in range(ppl)] + mx + [f"v_cmp_le_f32 vcc, %[thr], {top}"]
            val += [f"v_fma_f32 {A[q]}, {V[q]}, 0.5, 0.5 clamp" for q in range(ppl)]
            for n in range(max(len(val), len(sal))):             # VALU instructions leave scc alone (s_add_u32 -> s_addc_u32)
                if n < len(val):
                    e(val[n])
                if n < len(sal):
                    e(sal[n])
            e(f"s_cbranch_vccnz {lab(f'SPK{j}')}")
            e(f"{lab(f'R{j}')}:")
            for q in range(ppl):
                e(f"v_fma_f32 {V[q]}, {A[q]}, 2.0, -1.0")
            if j < EV_CHUNK - 2:
                e("s_setpc_b64 s[100:101]")
            elif j == EV_CHUNK - 2:
                # event 15 is all that is left: a step of its own, or the start of a step that goes on in the next chunk
                e(f"s_bitcmp1_b32 %[m], {EV_CHUNK - 1}")
                e(f"s_cbranch_scc1 {lab('SE15')}")
                for q in range(ppl):
                    e(f"v_mov_b32 {A[q]}, {W(EV_CHUNK - 1)[q]}")
                e(f"s_branch {lab('END')}")
            else:
                for q in range(ppl):
                    e(f"v_mov_b32 {A[q]}, 0")
                e(f"s_branch {lab('END')}")
            spike_path(c, lab, j, "", cnt)

    def spike_path(c, lab, i, m, cnt):
        c(f"{lab(f'SPK{i}{m}')}:")
        for k in range(ppl):
            c(f"v_max_f32 %[t], 0, {V[k]}")      # a NaN membrane (never a signalling one: it is a sum) gives 0
            c("v_floor_f32 %[t], %[t]")
            c(f"v_sub_f32 {V[k]}, {V[k]}, %[t]")
            c(f"v_add_f32 {C[k]}, {C[k]}, %[t]")
        c(f"s_bitcmp0_b32 {cnt(i)}, 0")
        c(f"s_cbranch_scc1 {lab(f'RC{i}{m}')}")
        if m == "H":                            # the flag exists only on counts below 2^23
            c(f"s_and_b32 %[sc], {cnt(i)}, 0x7ffffffe")
            c("s_cmp_lt_u32 %[sc], 0x4b000000")
            c(f"s_cbranch_scc0 {lab(f'RC{i}{m}')}")
        # dropped empty steps follow: a membrane left at exactly 1.0 by the clip fires there
        for k in range(ppl):
            c(f"v_fma_f32 %[t], {V[k]}, 0.5, 0.5 clamp")
            c("v_fma_f32 %[t], %[t], 2.0, -1.0")
            c(f"v_add_f32 %[t2], 1.0, {C[k]}")
            c("v_cmp_eq_f32 vcc, 1.0, %[t]")
            c("s_nop 1")
            c(f"v_cndmask_b32 {C[k]}, {C[k]}, %[t2], vcc")
            c(f"v_cndmask_b32_e64 {V[k]}, {V[k]}, 0, vcc")
        c(f"{lab(f'RC{i}{m}')}:")                # the membranes changed: redo the first half of the clip
        for k in range(ppl):
            c(f"v_fma_f32 {A[k]}, {V[k]}, 0.5, 0.5 clamp")
        c(f"s_branch {lab(f'R{i}{m}')}")

    def chunk(buf, other, name):
        e = hot.append
        off = lambda i: f"s{buf + 2 * i}"
        cnt = lambda i: f"s{buf + 2 * i + 1}"
        lab = lambda n: f"{n}{name}_%="

        # ---- weight reads, next chunk's prefetch ----
        # Every weight register first holds the address of its own read (a read may overwrite its address register:
        # the address is consumed when the read issues), so all addresses are computed before the first read is
        # issued and no read waits for the add right before it.  Same-box, against the address right before its
        # read: groups of two events -1.9 %, of four, eight or all sixteen -2.8 %.
        def read_phase(e, off0):      # off0: register that holds the first event's offset without the EV_CHUNK_MIXED flag
            for i0 in range(0, EV_CHUNK, READ_GROUP):
                ev = list(range(i0, min(EV_CHUNK, i0 + READ_GROUP)))
                for i in ev:
                    e(f"v_add_u32 {W(i)[0]}, {off0 if i == 0 else off(i)}, %[lane]")
                for i in ev:
                    if ppl == 3:
                        # The third place's weights sit in a plane of their own at LDS address 0, rows of 256 bytes = half
                        # the row stride the event word is scaled by, and are read by M0 + 4 * lane: a scalar shift in
                        # place of a second vector address.  (A write of M0 needs one other instruction before the read
                        # that uses it: the 8-byte read.)  Microbenchmark of the whole mix: -5 %.
                        e(f"s_lshr_b32 m0, {off0 if i == 0 else off(i)}, 1")
                    for ln in read(i, W(i)[0]):
                        if PROBE != "nolds":
                            e(ln)
            e("s_add_u32 %[soff], %[soff], 0x80")
            e("s_sub_u32 %[nch], %[nch], 1")          # borrow (scc) = no chunk follows; [nch] = chunks after the next one
            # LDS and scalar memory share one counter and scalar loads return out of order: the reads must have landed
            # before the prefetch is issued, or waiting for them would wait for the prefetch as well
            if PROBE != "nowait":
                e("s_waitcnt lgkmcnt(0)")

        # The chunk's flag first: a chunk whose counts are all 1 (96 % at the benchmark's density) uses its first offset
        # word as it is; the others get a copy of the read phase that masks the flag off, behind the loop.
        e(f"s_bitcmp1_b32 {off(0)}, 31")
        e(f"s_cbranch_scc1 {lab('MIXED')}")
        if JUMP:
            # bit i of [m] = event i is the LAST of its step, from the chunk's count words as one lane-parallel compare
            # (lane i holds count word i & 15; loaded one chunk ahead); bit 16 = sentinel "the step goes on in the next chunk"
            e("s_waitcnt vmcnt(0)")
            e("v_cmp_gt_i32 vcc, 0, %[vw]")
            e("s_and_b32 %[m], vcc_lo, 0xffff")
            e("s_or_b32 %[m], %[m], 0x10000")
        read_phase(e, off(0))
        e(f"s_cbranch_scc1 {lab('A0')}")
        e(f"s_load_dwordx16 s[{other}:{other + 15}], %[ev], %[soff] offset:0x0")
        e(f"s_load_dwordx16 s[{other + 16}:{other + 31}], %[ev], %[soff] offset:0x40")
        if JUMP:
            e("v_add_u32 %[va], %[soff], %[voff]")
            e("global_load_dword %[vw], %[va], %[ev]")
        # A chunk with counts other than 1: is any of them 2^23 or more (mantissa LSB = data, not the EV_EMPTY_NEXT
        # flag)?  The OR of the 16 count words is below 0x4b000000 only if every count is (a few combinations of
        # counts from 2^21 up test positive as well and take the careful variant, which is correct for any chunk).
        c = warm.append
        c(f"{lab('MIXED')}:")
        c(f"s_and_b32 %[so], {off(0)}, 0x7fffffff")
        read_phase(c, "%[so]")
        c(f"s_cbranch_scc1 {lab('NOPF')}")
        c(f"s_load_dwordx16 s[{other}:{other + 15}], %[ev], %[soff] offset:0x0")
        c(f"s_load_dwordx16 s[{other + 16}:{other + 31}], %[ev], %[soff] offset:0x40")
        if JUMP:
            c("v_add_u32 %[va], %[soff], %[voff]")
            c("global_load_dword %[vw], %[va], %[ev]")       # overwrites this chunk's (unused) count words: loads return in order
        c(f"{lab('NOPF')}:")
        c(f"s_or_b32 %[sc], {cnt(0)}, {cnt(1)}")
        for i in range(2, EV_CHUNK):
            c(f"s_or_b32 %[sc], %[sc], {cnt(i)}")
        c("s_and_b32 %[sc], %[sc], 0x7fffffff")
        c("s_cmp_lt_u32 %[sc], 0x4b000000")
        c(f"s_cbranch_scc1 {lab('A0M')}")
        c(f"s_branch {lab('A0H')}")

        main = hot.append
        for m in ("", "M", "H"):     # counts all 1 / all below 2^23 / any
            mixed = m != ""
            # the accumulate chain of a chunk whose counts are all 1 stays in the loop body and falls through into the
            # chunk's tail (-0.6 % same-box against a branch to it); everything else sits behind the loop
            e = warm.append if mixed else main
            al = lambda i: lab(f"A{i}{m}") if i < EV_CHUNK else lab("END")
            if JUMP and not mixed:
                jump_chunk(main, cold.append, lab, W, cnt)
                main(f"{lab('END')}:")
                main("s_waitcnt lgkmcnt(0)")          # the next chunk's words
                main("s_cmp_eq_u32 %[nch], -1")          # wrapped around: that was the last chunk
                continue

            def count_to_sc(i, back):      # [sc] = count of event i as a float (flag bits cleared)
                e(f"s_and_b32 %[sc], {cnt(i)}, 0x7ffffffe")
                if m == "H":
                    e("s_cmp_lt_u32 %[sc], 0x4b000000")
                    e(f"s_cbranch_scc0 {lab(f'BIG{back}')}")
                    e(f"{lab(f'BACK{back}')}:")
                    cold.append(f"{lab(f'BIG{back}')}:")      # count >= 2^23: its mantissa LSB is data, not a flag
                    cold.append(f"s_and_b32 %[sc], {cnt(i)}, 0x7fffffff")
                    cold.append(f"s_branch {lab(f'BACK{back}')}")

            # ---- chain A ----
            for i in range(EV_CHUNK):
                e(f"{al(i)}:")
                w = W(i)
                if mixed:
                    count_to_sc(i, f"A{i}")
                    op = lambda k: f"v_fmac_f32 {A[k]}, %[sc], {w[k]}"
                else:
                    op = lambda k: f"v_add_f32 {A[k]}, {A[k]}, {w[k]}"
                e(op(0))
                e(f"s_cmp_gt_i32 {cnt(i)}, -1")
                for k in range(1, ppl):
                    e(op(k))
                e(f"s_cbranch_scc0 {lab(f'S{i}{m}')}")
            if mixed:
                e(f"s_branch {lab('END')}")
            else:
                main(f"{lab('END')}:")
                main("s_waitcnt lgkmcnt(0)")          # the next chunk's words
                main("s_cmp_eq_u32 %[nch], -1")          # wrapped around: that was the last chunk
            e = warm.append

            # ---- chain S ----
            pair_first = PAIR_FIRST and not mixed
            order = (list(range(0, EV_CHUNK, 2)) + list(range(1, EV_CHUNK, 2))) if pair_first else list(range(EV_CHUNK))
            for pos, i in enumerate(order):
                nxt = order[pos + 1] if pos + 1 < EV_CHUNK else None     # the block laid out after this one
                e(f"{lab(f'S{i}{m}')}:")
                for k in range(ppl):
                    e(f"v_add_f32 {V[k]}, {V[k]}, {A[k]}")
                e(f"{lab(f'SM{i}{m}')}:")                  # entry for a step of one event: its weights went straight to v
                mx, top = fire_max()
                for ln in mx:
                    e(ln)
                e(f"v_cmp_le_f32 vcc, %[thr], {top}")
                # first half of the clip into the (now dead) accumulator registers, between the compare and its branch
                for k in range(ppl):
                    e(f"v_fma_f32 {A[k]}, {V[k]}, 0.5, 0.5 clamp")
                e(f"s_cbranch_vccnz {lab(f'SPK{i}{m}')}")
                e(f"{lab(f'R{i}{m}')}:")
                if i + 1 < EV_CHUNK and mixed:
                    count_to_sc(i + 1, f"S{i}")
                if pair_first and i + 2 < EV_CHUNK:
                    # Events i+1 and i+2 open the next step together: acc = w[i+1] + w[i+2] (fl(fl(0 + a) + b) is
                    # fl(a + b)), one add per place instead of a move and an add.  A step of a single event (16 % at
                    # the benchmark's density) takes the move, out of line.
                    w1, w2 = W(i + 1), W(i + 2)
                    e(f"s_cmp_gt_i32 {cnt(i + 1)}, -1")
                    for k in range(ppl):
                        e(f"v_fma_f32 {V[k]}, {A[k]}, 2.0, -1.0")
                    e(f"s_cbranch_scc0 {lab(f'ONE{i}')}")
                    e(f"v_add_f32 {A[0]}, {w1[0]}, {w2[0]}")
                    e(f"s_cmp_gt_i32 {cnt(i + 2)}, -1")
                    for k in range(1, ppl):
                        e(f"v_add_f32 {A[k]}, {w1[k]}, {w2[k]}")
                    e(f"s_cbranch_scc1 {al(i + 3)}")
                    if nxt != i + 2:
                        e(f"s_branch {lab(f'S{i + 2}{m}')}")
                    cold.append(f"{lab(f'ONE{i}')}:")
                    for k in range(ppl):
                        cold.append(f"v_add_f32 {V[k]}, {V[k]}, {w1[k]}")
                    cold.append(f"s_branch {lab(f'SM{i + 1}{m}')}")
                else:
                    for k in range(ppl):
                        e(f"v_fma_f32 {V[k]}, {A[k]}, 2.0, -1.0")
                    if i + 1 < EV_CHUNK:
                        w = W(i + 1)
                        e(f"s_cmp_gt_i32 {cnt(i + 1)}, -1")
                        for k in range(ppl):
                            e(f"v_mul_f32 {A[k]}, %[sc], {w[k]}" if mixed else f"v_mov_b32 {A[k]}, {w[k]}")
                        e(f"s_cbranch_scc1 {al(i + 2)}")
                        if nxt != i + 1:
                            e(f"s_branch {lab(f'S{i + 1}{m}')}")
                    else:
                        for k in range(ppl):
                            e(f"v_mov_b32 {A[k]}, 0")       # the step's successor is in the next chunk
                        e(f"s_branch {lab('END')}")

                spike_path(cold.append, lab, i, m, cnt)


    e = hot.append
    for k in range(ppl):
        e(f"v_mov_b32 {A[k]}, 0")
    # the stream's length travels with its first chunk: one memory latency per query instead of two in a row
    e("s_load_dword %[nch], %[nwp], 0x0")
    e(f"s_load_dwordx16 s[{EA}:{EA + 15}], %[ev], %[soff] offset:0x0")
    e(f"s_load_dwordx16 s[{EA + 16}:{EA + 31}], %[ev], %[soff] offset:0x40")
    if JUMP:
        e("global_load_dword %[vw], %[voff], %[ev]")     # the first chunk's count words, lane i: word 2 * (i & 15) + 1
        e("s_getpc_b64 s[100:101]")
        e("ANCHOR_%=:")
        e("s_mov_b32 %[bl], s100")
        e("s_mov_b32 %[bh], s101")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_lshr_b32 %[nch], %[nch], {EV_CHUNK.bit_length()}")     # words -> chunks of 2 * EV_CHUNK words
    e("s_cmp_eq_u32 %[nch], 0")
    e("s_cbranch_scc1 DONE_%=")
    e("s_sub_u32 %[nch], %[nch], 1")                 # chunks after the first
    e("LOOP_%=:")
    chunk(EA, EB, "a")
    e("s_cbranch_scc1 DONE_%=")
    chunk(EB, EA, "b")
    e("s_cbranch_scc0 LOOP_%=")
    e("s_branch DONE_%=")
    tail = ["DONE_%=:"] + (["s_waitcnt vmcnt(0)"] if JUMP else [])
    return hot + warm + cold + tail


def main():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "lens_amd", "csrc")
    for ppl in (1, 2, 3, 4):
        path = os.path.join(out_dir, f"output_loop_p{ppl}.inc")
        lines = gen(ppl)
        with open(path, "w") as f:
            f.write(f"// GENERATED by scripts/gen_output_loop.py -- do not edit.  gfx950 chunk loop of "
                    f"k_output_counts<{ppl}, true>.\n")
            for ln in lines:
                f.write('"' + ln + '\\n"\n')
        print(path, len(lines), "lines")


if __name__ == "__main__":
    main()

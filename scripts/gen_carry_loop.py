#!/usr/bin/env python3
"""Generates lens_amd/csrc/carry_l2_loop.inc: gfx950 assembly of k_output_carry_l2's chunk loop -- one query of a carried
stream against ONE wavefront's 64 places (one per lane), the weights read through L2.  The generated file is
committed; rerun this script after editing.

Why: with carried state a stream is sequential in the query index and every 64-place tile is walked by a lone
wavefront, which runs at the LATENCY of its own chain.  Read through L2 (no LDS tile, so every tile of a large database
is resident at once) a chunk's 16 weight rows take ~0.5 us to arrive.  Here the weights of DEPTH chunks are in flight
while a chunk is integrated:
    iteration j:  wait for the stream words of chunk j (vector load, 4 iterations old) -> LAST mask by one compare
                  wait for the scalar copy of chunk j + 3 -> issue its 16 row loads into ring slot (j + 3) % 4
                  request the scalar copy of chunk j + 4 (other SGPR buffer)
                  wait for the rows of chunk j (3 iterations old) -> integrate chunk j (chains A / S as in
                  gen_output_loop.py, the LAST test on the mask instead of the count word's sign)
                  request the stream words of chunk j + 4 (vector load into the register chunk j's words leave)
The scalar copy (s_load_dwordx16 x 2) is only needed to ISSUE the row loads (offsets in SGPRs for free), so two SGPR
buffers serve any depth; flags and counts at integration time come from the vector copy of the words (lane l holds
word l & 31: v_cmp for the LAST mask, v_readlane for a count or the EMPTY_NEXT flag on the rare paths).
vmcnt discipline: loads return in order and every iteration issues exactly 16 + 1 vector loads -- chunk indices past the
end of the stream are CLAMPED to its last chunk (redundant, harmless reads of valid addresses), never skipped -- so the
waits are constants: 51 younger loads behind the words of chunk j, 51 behind its rows.
Same f32 operations in the same order as integrate_chunk in lens_kernels.hip: bit-identical.

Operands: [v0] membrane (in/out), [c0] spike count (out), temporaries [a0] [t] [t2] [va] (VGPR) and [w0]..[w3] (VGPR: the
ring of stream words), scalars [nch] [soff] [sl] [sn] [sc] [so] [cw] [m] [k]; inputs [ev] stream pointer (SGPR pair),
[nwp] address of the stream's length in words, [wb] SGPR pair: address of row 0, place 0 of this tile in W2^T, [lane4]
4 x lane, [voff] 4 x (lane & 31), [thr] 1 - 2^-24 (VGPR).  The streams are written with the byte stride of a row of
W2^T as their event scale (lens_event_scale for this layout), so an offset word needs no multiply.
Clobbers: s[36:99] (two chunk buffers), v[64:127] (the ring of rows), vcc, scc.
"""
import os

EA, EB = 36, 68
EV_CHUNK = 16
DEPTH = 4
WBASE = 64
WAIT = 3 * (EV_CHUNK + 1)       # vector loads younger than the ones an iteration waits for


def W(slot, i):
    return f"v{WBASE + EV_CHUNK * slot + i}"


def gen():
    hot, warm, cold = [], [], []
    V, C, A = "%[v0]", "%[c0]", "%[a0]"

    def issue(e, buf, slot):
        """row loads of the chunk in SGPR buffer `buf` into ring slot `slot`: all addresses first (each in the register
        its load will fill), then the loads"""
        off = lambda i: f"s{buf + 2 * i}"
        e(f"s_and_b32 %[sc], {off(0)}, 0x7fffffff")          # EV_CHUNK_MIXED off
        for i in range(EV_CHUNK):                             # an event's offset word is the byte offset of its row of W2^T
            e(f"v_add_u32 {W(slot, i)}, {'%[sc]' if i == 0 else off(i)}, %[lane4]")
        for i in range(EV_CHUNK):
            e(f"global_load_dword {W(slot, i)}, {W(slot, i)}, %[wb]")

    def request(e, buf):
        e(f"s_load_dwordx16 s[{buf}:{buf + 15}], %[ev], %[sn] offset:0x0")
        e(f"s_load_dwordx16 s[{buf + 16}:{buf + 31}], %[ev], %[sn] offset:0x40")

    def words(e, reg):
        e("v_add_u32 %[va], %[sn], %[voff]")
        e(f"global_load_dword {reg}, %[va], %[ev]")

    def integrate(slot, wreg, name):
        """chains A / S for the chunk in ring slot `slot`, stream words in `wreg`; falls out at END<name>"""
        lab = lambda n: f"{n}{name}_%="
        not_last = lambda i: f"s_bitcmp0_b32 %[m], {2 * i + 1}"           # scc = 1: event i does not end its step
        w = lambda i: W(slot, i)
        main = hot.append
        main(f"s_bitcmp1_b32 %[so], 31")
        main(f"s_cbranch_scc1 {lab('A0H')}")
        for m in ("", "H"):          # counts all 1 / any (a count of 2^23 or more carries data in its mantissa LSB)
            mixed = m != ""
            e = warm.append if mixed else main
            al = lambda i: lab(f"A{i}{m}") if i < EV_CHUNK else lab("END")

            def count_to_sc(i, back):      # [sc] = count of event i as a float (flag bits cleared)
                e(f"v_readlane_b32 %[cw], {wreg}, {2 * i + 1}")
                e("s_and_b32 %[sc], %[cw], 0x7ffffffe")
                e("s_cmp_lt_u32 %[sc], 0x4b000000")
                e(f"s_cbranch_scc0 {lab(f'BIG{back}')}")
                e(f"{lab(f'BACK{back}')}:")
                cold.append(f"{lab(f'BIG{back}')}:")
                cold.append("s_and_b32 %[sc], %[cw], 0x7fffffff")
                cold.append(f"s_branch {lab(f'BACK{back}')}")

            # ---- chain A: event i joins the running step ----
            for i in range(EV_CHUNK):
                e(f"{al(i)}:")
                if mixed:
                    count_to_sc(i, f"A{i}")
                    e(f"v_fmac_f32 {A}, %[sc], {w(i)}")
                else:
                    e(f"v_add_f32 {A}, {A}, {w(i)}")
                e(not_last(i))
                e(f"s_cbranch_scc0 {lab(f'S{i}{m}')}")
            if mixed:
                e(f"s_branch {lab('END')}")
            else:
                main(f"{lab('END')}:")
            e = warm.append

            # ---- chain S: event i ends its step; events i+1 (and i+2) open the next one ----
            pair_first = not mixed
            order = (list(range(0, EV_CHUNK, 2)) + list(range(1, EV_CHUNK, 2))) if pair_first else list(range(EV_CHUNK))
            for pos, i in enumerate(order):
                nxt = order[pos + 1] if pos + 1 < EV_CHUNK else None
                e(f"{lab(f'S{i}{m}')}:")
                e(f"v_add_f32 {V}, {V}, {A}")
                e(f"{lab(f'SM{i}{m}')}:")
                e(f"v_cmp_le_f32 vcc, %[thr], {V}")
                e(f"v_fma_f32 {A}, {V}, 0.5, 0.5 clamp")
                e(f"s_cbranch_vccnz {lab(f'SPK{i}{m}')}")
                e(f"{lab(f'R{i}{m}')}:")
                if i + 1 < EV_CHUNK and mixed:
                    count_to_sc(i + 1, f"S{i}")
                if pair_first and i + 2 < EV_CHUNK:
                    e(not_last(i + 1))
                    e(f"v_fma_f32 {V}, {A}, 2.0, -1.0")
                    e(f"s_cbranch_scc0 {lab(f'ONE{i}')}")
                    e(f"v_add_f32 {A}, {w(i + 1)}, {w(i + 2)}")
                    e(not_last(i + 2))
                    e(f"s_cbranch_scc1 {al(i + 3)}")
                    if nxt != i + 2:
                        e(f"s_branch {lab(f'S{i + 2}{m}')}")
                    cold.append(f"{lab(f'ONE{i}')}:")
                    cold.append(f"v_add_f32 {V}, {V}, {w(i + 1)}")
                    cold.append(f"s_branch {lab(f'SM{i + 1}{m}')}")
                else:
                    e(f"v_fma_f32 {V}, {A}, 2.0, -1.0")
                    if i + 1 < EV_CHUNK:
                        e(not_last(i + 1))
                        e(f"v_mul_f32 {A}, %[sc], {w(i + 1)}" if mixed else f"v_mov_b32 {A}, {w(i + 1)}")
                        e(f"s_cbranch_scc1 {al(i + 2)}")
                        if nxt != i + 1:
                            e(f"s_branch {lab(f'S{i + 1}{m}')}")
                    else:
                        e(f"v_mov_b32 {A}, 0")
                        e(f"s_branch {lab('END')}")

                # spike path of step i
                c = cold.append
                c(f"{lab(f'SPK{i}{m}')}:")
                c(f"v_max_f32 %[t], 0, {V}")
                c("v_floor_f32 %[t], %[t]")
                c(f"v_sub_f32 {V}, {V}, %[t]")
                c(f"v_add_f32 {C}, {C}, %[t]")
                c(f"v_readlane_b32 %[cw], {wreg}, {2 * i + 1}")
                c("s_bitcmp0_b32 %[cw], 0")
                c(f"s_cbranch_scc1 {lab(f'RC{i}{m}')}")
                if mixed:                           # the flag exists only on counts below 2^23
                    c("s_and_b32 %[cw], %[cw], 0x7ffffffe")
                    c("s_cmp_lt_u32 %[cw], 0x4b000000")
                    c(f"s_cbranch_scc0 {lab(f'RC{i}{m}')}")
                c(f"v_fma_f32 %[t], {V}, 0.5, 0.5 clamp")
                c("v_fma_f32 %[t], %[t], 2.0, -1.0")
                c(f"v_add_f32 %[t2], 1.0, {C}")
                c("v_cmp_eq_f32 vcc, 1.0, %[t]")
                c("s_nop 1")
                c(f"v_cndmask_b32 {C}, {C}, %[t2], vcc")
                c(f"v_cndmask_b32_e64 {V}, {V}, 0, vcc")
                c(f"{lab(f'RC{i}{m}')}:")
                c(f"v_fma_f32 {A}, {V}, 0.5, 0.5 clamp")
                c(f"s_branch {lab(f'R{i}{m}')}")

    e = hot.append
    wr = [f"%[w{s}]" for s in range(DEPTH)]
    e(f"v_mov_b32 {A}, 0")
    e(f"v_mov_b32 {C}, 0")
    e("s_load_dword %[nch], %[nwp], 0x0")
    e(f"s_load_dwordx16 s[{EA}:{EA + 15}], %[ev], 0x0")
    e(f"s_load_dwordx16 s[{EA + 16}:{EA + 31}], %[ev], 0x40")
    e(f"global_load_dword {wr[0]}, %[voff], %[ev]")                    # words of chunk 0
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_lshr_b32 %[nch], %[nch], {EV_CHUNK.bit_length()}")         # words -> chunks
    e("s_cmp_eq_u32 %[nch], 0")
    e("s_cbranch_scc1 DONE_%=")
    e("s_sub_u32 %[sl], %[nch], 1")
    e("s_lshl_b32 %[sl], %[sl], 7")                                   # byte offset of the stream's last chunk
    issue(e, EA, 0)
    bufs = [EA, EB]
    for c in range(1, DEPTH):                                        # chunks 1 .. DEPTH-1 (clamped to the last chunk)
        e(f"s_min_u32 %[sn], {hex(128 * c)}, %[sl]")
        request(e, bufs[c % 2])
        words(e, wr[c])
        if c < DEPTH - 1:
            e("s_waitcnt lgkmcnt(0)")
            issue(e, bufs[c % 2], c)
    e(f"s_mov_b32 %[soff], {hex(128 * (DEPTH - 1))}")               # unclamped offset of the chunk requested last
    e("s_mov_b32 %[k], 0")
    e("LOOP_%=:")
    for it in range(DEPTH):
        name = f"i{it}"
        e(f"s_cmp_ge_u32 %[k], %[nch]")
        e("s_cbranch_scc1 DONE_%=")
        e(f"s_waitcnt vmcnt({WAIT})")                                 # the words of this chunk
        e(f"v_cmp_gt_i32 vcc, 0, {wr[it]}")
        e("s_mov_b32 %[m], vcc_lo")                                  # bit 2i + 1: event i is the LAST of its step
        e(f"v_readfirstlane_b32 %[so], {wr[it]}")                    # bit 31: EV_CHUNK_MIXED
        e("s_waitcnt lgkmcnt(0)")                                    # the scalar copy of chunk k + DEPTH - 1
        issue(e, bufs[(it + DEPTH - 1) % 2], (it + DEPTH - 1) % DEPTH)
        e("s_add_u32 %[soff], %[soff], 0x80")
        e("s_min_u32 %[sn], %[soff], %[sl]")
        request(e, bufs[(it + DEPTH) % 2])
        e(f"s_waitcnt vmcnt({WAIT})")                                 # the rows of this chunk
        integrate(it, wr[it], name)
        words(e, wr[it])                                             # words of chunk k + DEPTH
        e("s_add_u32 %[k], %[k], 1")
    e("s_branch LOOP_%=")
    return hot + warm + cold + ["DONE_%=:", "s_waitcnt vmcnt(0) lgkmcnt(0)"]


def main():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "lens_amd", "csrc")
    path = os.path.join(out_dir, "carry_l2_loop.inc")
    lines = gen()
    with open(path, "w") as f:
        f.write("// GENERATED by scripts/gen_carry_loop.py -- do not edit.  gfx950 chunk loop of k_output_carry_l2.\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    print(path, len(lines), "lines")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates gpu-homomorphic-encryption_amd/csrc/wide_asm.inc: the hand-scheduled AMDGCN blocks of the full-width Montgomery
product and butterfly (ntt_wide.hip.h).  Run from the repo root:  python3 scripts/gen_wide_asm.py

Why generated text: every block is ONE inline-asm statement (hipcc pads each statement boundary with wait states and cannot
see inside), and inside a block the carry-outs of v_mad_u64_u32 / v_add_co_u32 travel through SGPR pairs, which on gfx950 need
two wait states between the VALU that writes them and the VALU that reads them.  The blocks are software-pipelined so that
independent instructions fill those slots (no s_nop except in the 1- and 2-product columns):

  macn_<C>      : one column of the product-scanning Montgomery product: (hi:lo) += sum of C 32x32 products; the carry of
                  product i is folded into `hi` two instructions after its v_mad (three SGPR pairs rotate: s[20:25]).
  macn_e_<C>    : the same plus one step of the final t - q borrow chain (the words of t become final one per column, so the
                  conditional subtraction costs no wait states of its own); the borrow lives in an "s" operand between blocks.
  wsel_<NW>     : the last two steps of that chain and the select t >= q ? t - q : t.
  wmontc_<NW> / wmontl_<NW> (round 3): the WHOLE Montgomery product as one block -- canonical result, or "lazy" (no closing subtraction:
                  a < 2q, b < q < 2^(32 NW - 2) in, t < 2q out).  Inside one block the columns alternate between two accumulator pairs
                  (fixed registers v[16:17] / v[18:19], carry word v20), the fold of a column's last carry writes the NEXT accumulator's high
                  word directly, m_k = column * qinv and its product m_k q_0 sit inside the column, and the wait states every carry needs
                  are filled by the neighbouring folds and the shift: no s_nop, no statement boundary, one v_mov per column less.
                  302 instead of ~340 instructions per product at NW = 8 (284 lazy).  whole_mont() below builds the instruction list,
                  check_hazards() proves the wait-state distances, run() interprets it (tests/test_wide_asm.py checks both on the CPU).
  waddsub_<NW>  : butterfly tail: (a, t) <- (a + t mod q, a - t mod q) as four interleaved carry chains a + t, a - t,
                  (a + t) - q, (a - t) + q (each chain's next link is three instructions after the previous one) and two selects.
"""
import os
import random

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-homomorphic-encryption_amd", "csrc", "wide_asm.inc")
SP = ["s[20:21]", "s[22:23]", "s[24:25]"]


def mac_lines(cnt, extra=None):
    """mads into %0 (64-bit), carries folded into %1; operands x_i = %(2+2i), y_i = %(3+2i).  extra = an instruction text placed
    after the first mad (the borrow-chain step of macn_e)."""
    out, pending = [], []
    for i in range(cnt):
        out.append(f"v_mad_u64_u32 %0, {SP[i % 3]}, %{2 + 2 * i}, %{3 + 2 * i}, %0")
        if i == 0 and extra:
            out.append(extra)
        pending.append(i)
        if i >= 2:
            j = pending.pop(0)
            out.append(f"v_addc_co_u32_e64 %1, {SP[j % 3]}, 0, %1, {SP[j % 3]}")
    if cnt == 1 and not extra:
        out.append("s_nop 1")
    elif cnt == 1:
        out.append("s_nop 0")
    elif cnt == 2 and not extra:
        out.append("s_nop 0")
    for j in pending:
        out.append(f"v_addc_co_u32_e64 %1, {SP[j % 3]}, 0, %1, {SP[j % 3]}")
    return out


def asm_block(lines, outs, ins, clobbers, indent="    "):
    body = "\n".join(f'{indent}    "{l}\\n\\t"' for l in lines)
    return (f"{indent}asm(\n{body}\n{indent}    : {', '.join(outs)}\n{indent}    : {', '.join(ins) if ins else ''}\n"
            f"{indent}    : {', '.join(chr(34) + c + chr(34) for c in clobbers)});\n")


def gen_macn(cnt):
    args = ", ".join(f"uint32_t x{i}, uint32_t y{i}" for i in range(cnt))
    ins = [v for i in range(cnt) for v in (f'"v"(x{i})', f'"v"(y{i})')]
    s = f"__device__ __forceinline__ void macn_{cnt}(uint64_t &lo, uint32_t &hi, {args}) {{\n"
    s += asm_block(mac_lines(cnt), ['"+v"(lo)', '"+v"(hi)'], ins, ["s20", "s21", "s22", "s23", "s24", "s25"])
    return s + "}\n"


def gen_macn_e(cnt):
    # operands: %0 lo, %1 hi, x/y pairs at %2.., then bor = %(2+2cnt) ("+s"), tq = %(3+2cnt) ("=&v" -- listed with the outputs), t, q
    # clang numbers outputs first: outputs lo, hi, bor, tq -> %0 %1 %2 %3; inputs start at %4
    args = ", ".join(f"uint32_t x{i}, uint32_t y{i}" for i in range(cnt))
    out = ""
    for first in (True, False):
        nm = f"macn_e{'0' if first else ''}_{cnt}"
        base = 4
        t, q = f"%{base + 2 * cnt}", f"%{base + 2 * cnt + 1}"
        extra = f"v_sub_co_u32_e64 %3, %2, {t}, {q}" if first else f"v_subb_co_u32_e64 %3, %2, {t}, {q}, %2"
        lines = []
        pending = []
        at = 1 if cnt >= 2 else 0                      # the borrow-chain step sits after this mad (keeps every carry two slots from its reader)
        for i in range(cnt):
            lines.append(f"v_mad_u64_u32 %0, {SP[i % 3]}, %{base + 2 * i}, %{base + 2 * i + 1}, %0")
            if i == at:
                lines.append(extra)
            pending.append(i)
            if i >= 2:
                j = pending.pop(0)
                lines.append(f"v_addc_co_u32_e64 %1, {SP[j % 3]}, 0, %1, {SP[j % 3]}")
        if cnt == 1:
            lines.append("s_nop 0")
        for j in pending:
            lines.append(f"v_addc_co_u32_e64 %1, {SP[j % 3]}, 0, %1, {SP[j % 3]}")
        ins = [v for i in range(cnt) for v in (f'"v"(x{i})', f'"v"(y{i})')] + ['"v"(t)', '"v"(q)']
        outs = ['"+v"(lo)', '"+v"(hi)', '"=s"(bor)' if first else '"+s"(bor)', '"=&v"(tq)']
        out += f"__device__ __forceinline__ void {nm}(uint64_t &lo, uint32_t &hi, uint64_t &bor, uint32_t &tq, uint32_t t, uint32_t q, {args}) {{\n"
        out += asm_block(lines, outs, ins, ["s20", "s21", "s22", "s23", "s24", "s25"])
        out += "}\n"
    return out


def gen_wsel(nw):
    """t[nw-2], t[nw-1] finish the borrow chain; then t[i] = borrow ? t[i] : tq[i] for every word."""
    # outputs: t0..t(nw-1) "+v" (%0..), tq(nw-2), tq(nw-1) "=&v", bor "+s"; inputs: tq0..tq(nw-3), q(nw-2), q(nw-1)
    T = lambda i: f"%{i}"
    TQH = lambda i: f"%{nw + (i - (nw - 2))}"          # tq[nw-2], tq[nw-1]
    BOR = f"%{nw + 2}"
    inb = nw + 3
    TQL = lambda i: f"%{inb + i}"                      # tq[0..nw-3]
    QH = lambda i: f"%{inb + (nw - 2) + (i - (nw - 2))}"
    lines = [f"v_subb_co_u32_e64 {TQH(nw - 2)}, {BOR}, {T(nw - 2)}, {QH(nw - 2)}, {BOR}", "s_nop 1",
             f"v_subb_co_u32_e64 {TQH(nw - 1)}, {BOR}, {T(nw - 1)}, {QH(nw - 1)}, {BOR}", "s_nop 1"]
    for i in range(nw):
        src = TQL(i) if i < nw - 2 else TQH(i)
        lines.append(f"v_cndmask_b32_e64 {T(i)}, {src}, {T(i)}, {BOR}")     # borrow set (t < q): keep t
    outs = [f'"+v"(t[{i}])' for i in range(nw)] + [f'"=&v"(tq[{nw - 2}])', f'"=&v"(tq[{nw - 1}])', '"+s"(bor)']
    ins = [f'"v"(tq[{i}])' for i in range(nw - 2)] + [f'"v"(q[{nw - 2}])', f'"v"(q[{nw - 1}])']
    s = f"__device__ __forceinline__ void wsel_{nw}(uint32_t (&t)[{nw}], uint32_t (&tq)[{nw}], const uint32_t (&q)[{nw}], uint64_t &bor) {{\n"
    s += asm_block(lines, outs, ins, [])
    return s + "}\n"


def gen_addsub(nw):
    # operands: a[i] = %i ("+v"), t[i] = %(nw+i) ("+v"), u[i] = %(2nw+i) ("=&v"), v[i] = %(3nw+i) ("=&v"), q[i] = %(4nw+i) ("v")
    A = lambda i: f"%{i}"
    Tt = lambda i: f"%{nw + i}"
    U = lambda i: f"%{2 * nw + i}"
    V = lambda i: f"%{3 * nw + i}"
    Q = lambda i: f"%{4 * nw + i}"
    cA, cB, cC, cD = "vcc", "s[20:21]", "s[22:23]", "s[24:25]"
    lines = []
    for i in range(nw):
        if i == 0:
            lines += [f"v_add_co_u32_e64 {U(i)}, {cA}, {A(i)}, {Tt(i)}",             # s = a + t
                      f"v_sub_co_u32_e64 {Tt(i)}, {cB}, {A(i)}, {Tt(i)}",            # d = a - t   (in place on t)
                      f"v_sub_co_u32_e64 {A(i)}, {cC}, {U(i)}, {Q(i)}",              # s - q       (in place on a)
                      f"v_add_co_u32_e64 {V(i)}, {cD}, {Tt(i)}, {Q(i)}"]             # d + q
        else:
            lines += [f"v_addc_co_u32_e64 {U(i)}, {cA}, {A(i)}, {Tt(i)}, {cA}",
                      f"v_subb_co_u32_e64 {Tt(i)}, {cB}, {A(i)}, {Tt(i)}, {cB}",
                      f"v_subb_co_u32_e64 {A(i)}, {cC}, {U(i)}, {Q(i)}, {cC}",
                      f"v_addc_co_u32_e64 {V(i)}, {cD}, {Tt(i)}, {Q(i)}, {cD}"]
    for i in range(nw):     # a - t borrowed: take d + q.  (the last write of s[20:21] is two instructions back)
        lines.append(f"v_cndmask_b32_e64 {Tt(i)}, {Tt(i)}, {V(i)}, {cB}")
    for i in range(nw):     # (a + t) - q borrowed: keep a + t
        lines.append(f"v_cndmask_b32_e64 {A(i)}, {A(i)}, {U(i)}, {cC}")
    outs = [f'"+v"(a[{i}])' for i in range(nw)] + [f'"+v"(t[{i}])' for i in range(nw)] + [f'"=&v"(u[{i}])' for i in range(nw)] + \
           [f'"=&v"(v[{i}])' for i in range(nw)]
    ins = [f'"v"(q[{i}])' for i in range(nw)]
    s = (f"// (a, t) <- (a + t mod q, a - t mod q) for a, t < q < 2^({32 * nw} - 1)\n"
         f"__device__ __forceinline__ void waddsub_{nw}(uint32_t (&a)[{nw}], uint32_t (&t)[{nw}], const uint32_t (&q)[{nw}]) {{\n"
         f"    uint32_t u[{nw}], v[{nw}];\n")
    s += asm_block(lines, outs, ins, ["vcc", "s20", "s21", "s22", "s23", "s24", "s25"])
    return s + "}\n"


# ---- the whole Montgomery product as ONE block ---------------------------------------------------------------------------------
A_LO, A_HI, B_LO, B_HI, HREG = "v16", "v17", "v18", "v19", "v20"
SP = ["s[20:21]", "s[22:23]", "s[24:25]"]
BOR = "s[26:27]"


def whole_mont(nw, lazy):
    """Instruction list (tuples) of t = a b R^-1 mod q.  Symbolic operands: ('a', i), ('b', i), ('q', i), ('m', i), ('t', i), ('tq', i), 'qinv',
    physical: A_LO.. ; accumulator pairs 'A' / 'B'."""
    ins = []
    g = 0                                                # global product counter: rotates the carry pairs
    pair = {"A": (A_LO, A_HI), "B": (B_LO, B_HI)}
    cur, nxt = "A", "B"
    first_acc = True                                     # the very first mad starts from the constant 0
    for K in range(2 * nw - 1):
        prods = [(("a", i), ("b", K - i)) for i in range(nw) if 0 <= K - i < nw]
        prods += [(("m", i), ("q", K - i)) for i in range(nw) if 0 <= K - i < nw and i < K]
        last_col = K == 2 * nw - 2
        pend = []                                        # carries not yet folded: (pair index, position of the mad)
        h_fresh = True                                   # the first fold of a column defines H
        def mad(x, y):
            nonlocal g, first_acc
            sp = SP[g % 3]; g += 1
            ins.append(("mad", cur, sp, x, y, "zero" if first_acc else cur)); first_acc = False
            pend.append(sp)
        def fold(dest_next=False):
            nonlocal h_fresh
            sp = pend.pop(0)
            if last_col:
                return                                   # the sum is below 2q < 2^(32 nw): nothing is carried out of the top pair
            dst = pair[nxt][1] if dest_next else HREG
            ins.append(("addc", dst, sp, "zero" if h_fresh else HREG, sp)); h_fresh = False
        n = len(prods)
        for idx, (x, y) in enumerate(prods):
            mad(x, y)
            if idx >= 2:
                fold()
        if K < nw:                                       # m_K = column * qinv; column += m_K q_0 (its low word becomes 0)
            ins.append(("mul_lo", ("m", K), pair[cur][0], "qinv"))
            mad(("m", K), ("q", 0))
            while len(pend) > 1:
                fold()
            ins.append(("mov", pair[nxt][0], pair[cur][1]))
            fold(dest_next=True)
        elif not last_col:
            if n == 2:
                ins.append(("mov", ("t", K - nw), pair[cur][0]))
                fold()
                ins.append(("mov", pair[nxt][0], pair[cur][1]))
                fold(dest_next=True)
            else:
                while len(pend) > 1:
                    fold()
                ins.append(("mov", ("t", K - nw), pair[cur][0]))
                ins.append(("mov", pair[nxt][0], pair[cur][1]))
                fold(dest_next=True)
        else:
            ins.append(("mov", ("t", nw - 2), pair[cur][0]))
            ins.append(("mov", ("t", nw - 1), pair[cur][1]))
        if not lazy and K >= nw + 1:                     # one step of t - q per column, as its word becomes final (t[K-nw-1] was written a column ago)
            w = K - nw - 1
            ins.insert(len(ins) - 2, ("sub" if w == 0 else "subb", ("tq", w), BOR, ("t", w), ("q", w)))
        cur, nxt = nxt, cur
    if not lazy:
        for w in (nw - 2, nw - 1):
            ins.append(("subb", ("tq", w), BOR, ("t", w), ("q", w)))
            ins.append(("nop", 1))
        for w in range(nw):
            ins.append(("cnd", ("t", w), ("tq", w), ("t", w), BOR))      # borrow set (t < q): keep t
    return ins


def check_hazards(ins):
    """An SGPR pair written by a VALU is read as carry-in / select no earlier than three instructions later (two wait states); nothing
    overwrites a pair whose value is still to be read."""
    last_write = {}
    for pos, i in enumerate(ins):
        op = i[0]
        reads, writes = [], []
        if op == "mad": writes = [i[2]]
        elif op == "addc": reads = [i[4]]; writes = [i[2]]
        elif op == "sub": writes = [i[2]]
        elif op == "subb": reads = [i[2]]; writes = [i[2]]
        elif op == "cnd": reads = [i[4]]
        for r in reads:
            wpos, width = last_write[r]
            gap = sum((x[1] + 1) if x[0] == "nop" else 1 for x in ins[wpos + 1:pos])
            assert gap >= 2, (pos, i, gap)
        for w in writes:
            last_write[w] = (pos, 0)
    # a carry pair must be folded before the next mad reuses it
    pending = {}
    for pos, i in enumerate(ins):
        if i[0] == "mad":
            assert i[2] not in pending, (pos, i)
            pending[i[2]] = pos
        elif i[0] == "addc":
            pending.pop(i[4], None)
    return True


def run(ins, nw, a, b, q, qinv):
    M = 0xFFFFFFFF
    reg = {"zero": 0, "qinv": qinv}
    for i in range(nw):
        reg[("a", i)] = (a >> (32 * i)) & M; reg[("b", i)] = (b >> (32 * i)) & M; reg[("q", i)] = (q >> (32 * i)) & M
    sg = {}
    pair = {"A": (A_LO, A_HI), "B": (B_LO, B_HI)}
    def rd64(p):
        if p == "zero": return 0
        lo, hi = pair[p]; return reg[lo] | (reg[hi] << 32)
    for i in ins:
        op = i[0]
        if op == "mad":
            _, d, sp, x, y, c = i
            v = reg[x] * reg[y] + rd64(c)
            sg[sp] = v >> 64
            lo, hi = pair[d]; reg[lo] = v & M; reg[hi] = (v >> 32) & M
        elif op == "addc":
            _, d, spo, x, spi = i
            v = reg[x] + sg[spi]
            assert v <= M
            reg[d] = v; sg[spo] = 0
        elif op == "mul_lo":
            reg[i[1]] = (reg[i[2]] * reg[i[3]]) & M
        elif op == "mov":
            reg[i[1]] = reg[i[2]]
        elif op in ("sub", "subb"):
            _, d, sp, x, y = i
            v = reg[x] - reg[y] - (sg[sp] if op == "subb" else 0)
            sg[sp] = 1 if v < 0 else 0
            reg[d] = v & M
        elif op == "cnd":
            _, d, x, y, sp = i
            reg[d] = reg[y] if sg[sp] else reg[x]
    return sum(reg[("t", i)] << (32 * i) for i in range(nw))



def gen_wmont(nw, lazy):
    """C++ wrapper + asm text of whole_mont(nw, lazy)."""
    ins = whole_mont(nw, lazy)
    check_hazards(ins)
    outs = [f'"=&v"(t[{i}])' for i in range(nw)] + [f'"=&v"(m[{i}])' for i in range(nw)]
    names = {("t", i): f"%{i}" for i in range(nw)}
    names.update({("m", i): f"%{nw + i}" for i in range(nw)})
    base = 2 * nw
    if not lazy:
        outs += [f'"=&v"(tq[{i}])' for i in range(nw)]
        names.update({("tq", i): f"%{base + i}" for i in range(nw)})
        base += nw
    inputs = []
    for arr in ("a", "b", "q"):
        for i in range(nw):
            names[(arr, i)] = f"%{base + len(inputs)}"
            inputs.append(f'"v"({arr}[{i}])')
    names["qinv"] = f"%{base + len(inputs)}"; inputs.append('"v"(qinv)')
    names["zero"] = "0"
    for r in (A_LO, A_HI, B_LO, B_HI, HREG):
        names[r] = r
    pr = {"A": "v[16:17]", "B": "v[18:19]"}
    N = lambda x: names[x]
    lines = []
    for i in ins:
        op = i[0]
        if op == "mad": lines.append(f"v_mad_u64_u32 {pr[i[1]]}, {i[2]}, {N(i[3])}, {N(i[4])}, {'0' if i[5] == 'zero' else pr[i[5]]}")
        elif op == "addc": lines.append(f"v_addc_co_u32_e64 {N(i[1])}, {i[2]}, 0, {N(i[3])}, {i[4]}")
        elif op == "mul_lo": lines.append(f"v_mul_lo_u32 {N(i[1])}, {N(i[2])}, {N(i[3])}")
        elif op == "mov": lines.append(f"v_mov_b32_e32 {N(i[1])}, {N(i[2])}")
        elif op == "sub": lines.append(f"v_sub_co_u32_e64 {N(i[1])}, {i[2]}, {N(i[3])}, {N(i[4])}")
        elif op == "subb": lines.append(f"v_subb_co_u32_e64 {N(i[1])}, {i[2]}, {N(i[3])}, {N(i[4])}, {i[2]}")
        elif op == "cnd": lines.append(f"v_cndmask_b32_e64 {N(i[1])}, {N(i[2])}, {N(i[3])}, {i[4]}")
        elif op == "nop": lines.append(f"s_nop {i[1]}")
    clob = ["v16", "v17", "v18", "v19", "v20", "s20", "s21", "s22", "s23", "s24", "s25"] + ([] if lazy else ["s26", "s27"])
    nm = f"wmont{'l' if lazy else 'c'}_{nw}"
    what = (f"t = a b 2^-{32 * nw} mod q for a < 2q, b < q < 2^{32 * nw - 2}: t < 2q (no closing subtraction)" if lazy else
            f"t = a b 2^-{32 * nw} mod q, canonical, for a, b < q < 2^{32 * nw - 1}")
    s = (f"// {what}; {len(ins)} instructions, one block\n"
         f"__device__ __forceinline__ void {nm}(uint32_t (&t)[{nw}], const uint32_t (&a)[{nw}], const uint32_t (&b)[{nw}], const uint32_t (&q)[{nw}], uint32_t qinv) {{\n"
         f"    uint32_t m[{nw}]{'' if lazy else f', tq[{nw}]'};\n")
    s += asm_block(lines, outs, inputs, clob)
    return s + "}\n"


def gen_free_tail(nw):
    """(a, t) <- (a + t, a - t + c) modulo 2^(32 nw), no reduction: the forward butterfly tail of the lazy class (c = 2q; the true values stay
    below 2^(32 nw), see ntt_wide.hip.h).  Three interleaved carry chains: d = a - t (into v), a += t (in place), t = d + c."""
    A = lambda i: f"%{i}"
    Tt = lambda i: f"%{nw + i}"
    V = lambda i: f"%{2 * nw + i}"
    C = lambda i: f"%{3 * nw + i}"
    cA, cB, cC = "s[20:21]", "s[22:23]", "s[24:25]"
    lines = []
    for i in range(nw):
        if i == 0:
            lines += [f"v_sub_co_u32_e64 {V(i)}, {cA}, {A(i)}, {Tt(i)}", f"v_add_co_u32_e64 {A(i)}, {cB}, {A(i)}, {Tt(i)}",
                      f"v_add_co_u32_e64 {Tt(i)}, {cC}, {V(i)}, {C(i)}"]
        else:
            lines += [f"v_subb_co_u32_e64 {V(i)}, {cA}, {A(i)}, {Tt(i)}, {cA}", f"v_addc_co_u32_e64 {A(i)}, {cB}, {A(i)}, {Tt(i)}, {cB}",
                      f"v_addc_co_u32_e64 {Tt(i)}, {cC}, {V(i)}, {C(i)}, {cC}"]
    outs = [f'"+v"(a[{i}])' for i in range(nw)] + [f'"+v"(t[{i}])' for i in range(nw)] + [f'"=&v"(v[{i}])' for i in range(nw)]
    ins = [f'"v"(c[{i}])' for i in range(nw)]
    s = (f"// (a, t) <- (a + t, a - t + c) mod 2^{32 * nw}: {len(lines)} instructions\n"
         f"__device__ __forceinline__ void wfree_{nw}(uint32_t (&a)[{nw}], uint32_t (&t)[{nw}], const uint32_t (&c)[{nw}]) {{\n    uint32_t v[{nw}];\n")
    s += asm_block(lines, outs, ins, ["s20", "s21", "s22", "s23", "s24", "s25"])
    return s + "}\n"


def gen_csub4(nw):
    """x_k <- x_k >= c ? x_k - c : x_k for four values at once (four interleaved borrow chains: no wait states), c < 2^(32 nw)."""
    SPK = ["s[20:21]", "s[22:23]", "s[24:25]", "s[26:27]"]
    X = lambda k, i: f"%{k * nw + i}"
    D = lambda k, i: f"%{4 * nw + k * nw + i}"
    C = lambda i: f"%{8 * nw + i}"
    lines = []
    for i in range(nw):
        for k in range(4):
            lines.append(f"v_sub_co_u32_e64 {D(k, i)}, {SPK[k]}, {X(k, i)}, {C(i)}" if i == 0 else
                         f"v_subb_co_u32_e64 {D(k, i)}, {SPK[k]}, {X(k, i)}, {C(i)}, {SPK[k]}")
    for k in range(4):
        for i in range(nw):
            lines.append(f"v_cndmask_b32_e64 {X(k, i)}, {D(k, i)}, {X(k, i)}, {SPK[k]}")      # borrow set (x < c): keep x
    outs = [f'"+v"(x{k}[{i}])' for k in range(4) for i in range(nw)] + [f'"=&v"(d{k}[{i}])' for k in range(4) for i in range(nw)]
    ins = [f'"v"(c[{i}])' for i in range(nw)]
    args = ", ".join(f"uint32_t (&x{k})[{nw}]" for k in range(4))
    s = (f"// four conditional subtractions of the same constant: {len(lines)} instructions\n"
         f"__device__ __forceinline__ void wcsub4_{nw}({args}, const uint32_t (&c)[{nw}]) {{\n    uint32_t d0[{nw}], d1[{nw}], d2[{nw}], d3[{nw}];\n")
    s += asm_block(lines, outs, ins, ["s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"])
    return s + "}\n"


def dispatcher(name, maxc, extra_params, extra_args):
    s = f"template <int CNT>\n__device__ __forceinline__ void {name}(uint64_t &lo, uint32_t &hi{extra_params}, const uint32_t (&x)[16], const uint32_t (&y)[16]) {{\n"
    for c in range(1, maxc + 1):
        args = ", ".join(f"x[{i}], y[{i}]" for i in range(c))
        s += f"    {'if' if c == 1 else 'else if'} constexpr (CNT == {c}) {name}_{c}(lo, hi{extra_args}, {args});\n"
    s += "}\n"
    return s


def main():
    parts = ["// GENERATED by scripts/gen_wide_asm.py -- do not edit by hand.  See that script for what the blocks are.\n#pragma once\n"]
    for c in range(1, 16):
        parts.append(gen_macn(c))
    for c in range(1, 15):
        parts.append(gen_macn_e(c))
    parts.append(dispatcher("macn", 15, "", ""))
    parts.append(dispatcher("macn_e", 14, ", uint64_t &bor, uint32_t &tq, uint32_t t, uint32_t q", ", bor, tq, t, q"))
    parts.append(dispatcher("macn_e0", 14, ", uint64_t &bor, uint32_t &tq, uint32_t t, uint32_t q", ", bor, tq, t, q"))
    for nw in (4, 8):
        parts.append(gen_wsel(nw))
        parts.append(gen_addsub(nw))
        parts.append(gen_wmont(nw, False))
        parts.append(gen_wmont(nw, True))
        parts.append(gen_free_tail(nw))
        parts.append(gen_csub4(nw))
    open(OUT, "w").write("\n".join(parts))
    print("wrote", OUT)


if __name__ == "__main__":
    main()

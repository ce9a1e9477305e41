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
  waddsub_<NW>  : butterfly tail: (a, t) <- (a + t mod q, a - t mod q) as four interleaved carry chains a + t, a - t,
                  (a + t) - q, (a - t) + q (each chain's next link is three instructions after the previous one) and two selects.
"""
import os

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
    open(OUT, "w").write("\n".join(parts))
    print("wrote", OUT)


if __name__ == "__main__":
    main()

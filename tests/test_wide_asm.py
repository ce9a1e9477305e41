"""CPU test of the generated AMDGCN blocks of the full-width class (scripts/gen_wide_asm.py -> csrc/wide_asm.inc): the whole Montgomery
product as one block is (1) interpreted instruction by instruction against Python big-integer arithmetic, (2) checked for the two wait states
gfx950 needs between a VALU that writes an SGPR carry and the VALU that reads it, and (3) the committed wide_asm.inc is what the generator emits."""
import importlib.util
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_wide_asm", os.path.join(ROOT, "scripts", "gen_wide_asm.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)


@pytest.mark.parametrize("nw", [4, 8])
@pytest.mark.parametrize("lazy", [False, True])
def test_whole_montgomery_block_is_the_montgomery_product(nw, lazy):
    ins = gen.whole_mont(nw, lazy)
    assert gen.check_hazards(ins)
    mads = sum(1 for i in ins if i[0] == "mad")
    assert mads == 2 * nw * nw                           # a_i b_j, m_i q_j (j >= 1) and m_k q_0
    assert not any(i[0] == "nop" for i in ins) or not lazy
    bits, rnd = 32 * nw, random.Random(1234 + nw)
    R = 1 << bits
    for trial in range(400):
        top = bits - 6 if lazy else bits - 1             # lazy class: six spare bits (ntt_wide.hip.h)
        qb = top if trial % 2 else rnd.randrange(bits - 33, top + 1)
        q = (rnd.getrandbits(qb) | 1 | (1 << (qb - 1))) if trial > 2 else (1 << top) - 1
        qinv = (-pow(q, -1, 1 << 32)) % (1 << 32)
        if lazy:                                         # forward butterflies feed values below 23 q; the pointwise product 23 q times 2 q
            a = rnd.randrange(23 * q) if trial % 3 else 23 * q - 1
            b = rnd.randrange(q) if trial % 4 else rnd.randrange(2 * q)
            if trial % 3 == 0 and trial % 4: b = q - 1
            bound = 2 * q
        else:
            a, b = (rnd.randrange(q), rnd.randrange(q)) if trial % 5 else (q - 1, q - 1)
            bound = q
        t = gen.run(ins, nw, a, b, q, qinv)
        assert (t * R - a * b) % q == 0
        assert t < bound, (trial, hex(q), hex(a), hex(b))


def test_committed_asm_is_what_the_generator_emits(tmp_path, monkeypatch):
    out = tmp_path / "wide_asm.inc"
    monkeypatch.setattr(gen, "OUT", str(out))
    gen.main()
    with open(os.path.join(ROOT, "gpu-homomorphic-encryption_amd", "csrc", "wide_asm.inc")) as f:
        assert f.read() == out.read_text(), "csrc/wide_asm.inc is stale: run python3 scripts/gen_wide_asm.py"

#!/usr/bin/env python3
"""Rewrites the packed-fp32 instructions of a gfx950 assembly file that MI355X computes wrongly next to matrix products.

The erratum (tools/ubench/pk_opsel_mfma.hip, profiles/r06_experiments.md "the sliced forward's heisenbug, decoded"):

    v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32   with   op_sel:[0,1]  or  op_sel:[0,1,x]

-- the LOW result takes src0's low register and src1's HIGH register -- reads that high register as ZERO in the lanes
48..63 now and then, whenever ANOTHER wave of the SIMD is running v_mfma.  Every other selection is right, in particular the
same product with its first two operands exchanged (op_sel:[1,0]).  Multiplication and addition commute and the two
factors of a fused multiply-add do, so the exchange is exact: same registers read, same result bits.

hipcc forms these instructions by itself (SLP vectoriser + the folding of the shuffles into op_sel); the library is built
through its assembly (csrc/Makefile) so that this script sees every kernel.  `--check` only reports (exit 1 if anything
is left to rewrite); tests/test_isa_lint.py runs the same scan over the disassembly of the built library.
"""
import re
import sys

INSN = re.compile(r"^(\s*)(v_pk_(?:mul|add|fma)_f32)\s+(.*?)\s*$")
MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")


def split_operands(text):
    """'v[2:3], v[4:5], 1.0 op_sel:[0,1] neg_lo:[1,0]' -> (['v[2:3]', 'v[4:5]', '1.0'], {'op_sel': [0, 1], ...}, order)"""
    mods, order = {}, []
    for m in MOD.finditer(text):
        mods[m.group(1)] = [int(x) for x in m.group(2).split(",")]
        order.append(m.group(1))
    ops_text = MOD.sub("", text).strip()
    ops, depth, cur = [], 0, ""
    for ch in ops_text:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops, mods, order


def dangerous(mods):
    sel = mods.get("op_sel")
    return sel is not None and len(sel) >= 2 and sel[0] == 0 and sel[1] == 1


def rewrite_line(line):
    """-> (new line, changed?)"""
    code, sep, comment = line.partition(";")
    m = INSN.match(code.rstrip("\n"))
    if not m:
        return line, False
    indent, op, rest = m.groups()
    ops, mods, order = split_operands(rest)
    if not dangerous(mods):
        return line, False
    # dst, src0, src1 (, src2): exchange the two factors / summands and the first two entries of every modifier
    ops[1], ops[2] = ops[2], ops[1]
    for k in mods:
        mods[k][0], mods[k][1] = mods[k][1], mods[k][0]
    text = indent + op + " " + ", ".join(ops)
    for k in order:
        text += " %s:[%s]" % (k, ",".join(str(x) for x in mods[k]))
    tail = "\n" if code.endswith("\n") or (not sep and line.endswith("\n")) else ""
    if sep:
        return text + " " + sep + comment, True
    return text + tail, True


def scan(lines):
    """-> [(line number, text)] of the dangerous instructions"""
    out = []
    for i, line in enumerate(lines, 1):
        code = line.partition(";")[0]
        m = INSN.match(code.rstrip("\n"))
        if m and dangerous(split_operands(m.group(3))[1]):
            out.append((i, code.strip()))
    return out


def main(argv):
    check = "--check" in argv
    files = [a for a in argv[1:] if not a.startswith("--")]
    if not files:
        print(__doc__)
        return 2
    left = 0
    for path in files:
        with open(path) as f:
            lines = f.readlines()
        if check:
            found = scan(lines)
            for i, text in found:
                print("%s:%d: %s" % (path, i, text))
            left += len(found)
            continue
        n = 0
        for i, line in enumerate(lines):
            lines[i], changed = rewrite_line(line)
            n += changed
        again = scan(lines)
        if again:
            print("%s: %d instruction(s) still dangerous after the rewrite, first: %s" % (path, len(again), again[0][1]))
            return 1
        if n:
            with open(path, "w") as f:
                f.writelines(lines)
        print("%s: %d packed fp32 instruction(s) rewritten" % (path, n))
    return 1 if left else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))

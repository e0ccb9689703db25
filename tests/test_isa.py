"""Build-time assertions on the generated gfx950 ISA of the sliced forward (csrc/msda_fwd_q8.hip) -- ADVICE r4 (medium).

Round 4 met an intermittent wrong result in that kernel under load (profiles/r04_experiments.md r04f-m) and held it off with
three properties of the instruction stream, none of which a compiler or ROCm update is obliged to keep:
  * its matrix-core products accumulate IN PLACE (destination = accumulator operand) in registers that are not the
    product's own A or B operand (inline assembly; tools/ubench/mfma_alias.hip, round 5, shows the overlap by itself is
    harmless on this part -- 1e9 products bit-equal -- so this is belt and braces, but it is what the fix shipped with);
  * the two K-blocks' offset reads are NOT merged into one ds_read2_b64 (the merge was in every failing build);
  * every product is fenced by wait states (s_nop) that the hazard recogniser cannot insert around inline assembly.
This test compiles the kernel to assembly (no GPU needed: hipcc cross-compiles) and asserts all three, next to the
full-size repeated-run stress test of the kernel on the GPU (tests/test_stress_gpu.py), which is what gates it as the default
for heads of 32 / 64 channels."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mm-interleaved_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-S", "--cuda-device-only"]


def _asm(tmp_path, unit):
    out = tmp_path / (unit + ".s")
    r = subprocess.run([HIPCC] + FLAGS + [os.path.join(CSRC, unit + ".hip"), "-o", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def _kernels(text, prefix):
    """{name: body} of the kernels whose mangled name contains ``prefix``."""
    res, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w*" + prefix + r"\w*):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line)
            if "s_endpgm" in line:
                res[name] = body
                name = None
    return res


def _regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r"^([va])(\d+)$", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None, set()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_sliced_forward_isa_keeps_what_the_fix_relies_on(tmp_path):
    text = _asm(tmp_path, "msda_fwd_q8")
    kernels = _kernels(text, "msda_fwd_q8")
    assert len(kernels) >= 2                                  # fp16 and bf16
    for name, body in kernels.items():
        code = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", "."))]
        mfma = [i for i, l in enumerate(code) if l.startswith("v_mfma")]
        assert len(mfma) >= 8, name
        for i in mfma:
            ops = code[i].split(None, 1)[1].split(",")
            dst, a, b, c = (_regs(t) for t in ops[:4])
            assert dst == c, f"{name}: a product does not accumulate in place: {code[i]}"
            assert not (dst[0] == a[0] and dst[1] & a[1]) and not (dst[0] == b[0] and dst[1] & b[1]), \
                f"{name}: a product's destination overlaps its own operand: {code[i]}"
            # wait states in front (>= 2: s_nop 1) unless the previous instruction is the chain's other product, and behind
            # the last product of a sequence (>= 8: s_nop 7) before anything else issues
            prev = code[i - 1]
            assert prev.startswith(("s_nop", "v_mfma")), f"{name}: no wait state in front of {code[i]} (after {prev})"
            nxt = code[i + 1]
            assert nxt.startswith(("s_nop", "v_mfma")), f"{name}: no wait state behind {code[i]} (before {nxt})"
        # the K-blocks' offset pairs stay two 8-byte reads: the record reads in front of the transposing reads
        assert not any(l.startswith("ds_read2_b64") for l in code), f"{name}: an LDS read pair was merged (ds_read2_b64)"
        assert any(l.startswith("ds_read_b64_tr_b16") for l in code), name

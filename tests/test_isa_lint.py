"""The built library must not contain the packed-fp32 instructions MI355X computes wrongly next to matrix products
(v_pk_{mul,add,fma}_f32 with op_sel:[0,1,..]: tools/ubench/pk_opsel_mfma.hip, tools/fix_pk_opsel.py).  The scan runs over the
disassembly of every code object embedded in libmmfs_msda.so -- the artefact that ships, not the intermediate files.

(Replaces round 5's tests/test_isa.py, which asserted three properties of msda_fwd_q8's instruction stream -- in-place inline-assembly
products, no ds_read2_b64, wait states around every product -- that round 4 had shipped as the fix of an intermittent wrong result.
Round 6 found the cause, profiles/r06_experiments.md r06aa: none of the three had anything to do with it.)"""
import os
import struct
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fix_pk_opsel  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "mm-interleaved_amd", "libmmfs_msda.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """The gfx950 code objects of a .hip_fatbin section (one bundle per translation unit)."""
    out, at = [], blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(MAGIC, at + len(MAGIC))
    return out


def test_rewriter_exchanges_the_commuting_operands():
    f = fix_pk_opsel.rewrite_line
    assert f("\tv_pk_mul_f32 v[54:55], v[50:51], v[64:65] op_sel:[0,1] op_sel_hi:[1,0]\n") == \
        ("\tv_pk_mul_f32 v[54:55], v[64:65], v[50:51] op_sel:[1,0] op_sel_hi:[0,1]\n", True)
    assert f("\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]\n") == \
        ("\tv_pk_add_f32 v[2:3], v[6:7], v[4:5] op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[1,0]\n", True)
    assert f("\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], -0.5 op_sel:[0,1,0] op_sel_hi:[1,0,0]\n") == \
        ("\tv_pk_fma_f32 v[2:3], v[6:7], v[4:5], -0.5 op_sel:[1,0,0] op_sel_hi:[0,1,0]\n", True)
    # what is right stays as it is: src0's high half, both high halves, selections of the HIGH result only
    for ok in ("\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[1,0,0]\n", "\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,1]\n",
               "\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[0,1]\n", "\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,0,1]\n",
               "\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n", "\tv_pk_mul_f16 v2, v4, v6 op_sel:[0,1]\n"):
        assert f(ok) == (ok, False)
    assert fix_pk_opsel.scan(["\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]\n", "\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0]\n"]) == \
        [(1, "v_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]")]


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))), reason="library / llvm-objdump not here")
def test_built_library_has_none_of_the_erratum_instructions(tmp_path):
    sec = tmp_path / "fatbin"
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=%s" % sec, LIB, str(tmp_path / "unused.so")], check=True)
    objs = code_objects(sec.read_bytes())
    assert len(objs) >= 15, "every translation unit of csrc/ carries a gfx950 code object"
    packed = mfma = 0
    for i, co in enumerate(objs):
        path = tmp_path / ("co%d.hsaco" % i)
        path.write_bytes(co)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", str(path)], check=True, capture_output=True, text=True).stdout
        lines = [ln.split("//")[0] for ln in dis.splitlines()]
        bad = fix_pk_opsel.scan(lines)
        assert not bad, "code object %d: %d dangerous packed fp32 instruction(s), first: %s" % (i, len(bad), bad[0][1])
        packed += sum("v_pk_" in ln and "_f32" in ln for ln in lines)
        mfma += sum("v_mfma" in ln for ln in lines)
    assert packed > 1000 and mfma > 100, "the scan saw the kernels (packed fp32: %d, matrix products: %d)" % (packed, mfma)

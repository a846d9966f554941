"""Static guard for a gfx950 store-data hazard in the built library (CPU only: disassembles libnerf_hip.so).

Observed on MI355X: a `buffer_store_dwordx4` (more than 64 bits of data) whose scalar offset is an SGPR, followed
immediately by a vector-ALU instruction that overwrites one of its data registers, can store a mix of old and new
values once the memory pipeline is busy.  hipcc inserts the wait state only when the scalar offset is an immediate.
The kernels therefore keep such stores either on an immediate offset (csrc/mlp_device.h, the derivative tiles) or
have their data registers rewritten by LDS loads only (the workspace copies inside the MFMA loops).  A register
allocator decision could silently break the second arrangement; run-to-run equality tests would only catch it when
the corruption is random.  This test fails the build instead: no wide buffer store with a register scalar offset may
be followed within two instruction slots by a VALU (or MFMA) instruction that writes one of its data VGPRs."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "nerf-pytorch_amd", "lib", "libnerf_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_disassembly():
    """Disassembly text of every gfx950 code object bundled in the library's .hip_fatbin section."""
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(LIB) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or LLVM tools not present")
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([tools[0], "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        assert starts, "no offload bundle in .hip_fatbin"
        for i, s in enumerate(starts):
            piece = os.path.join(tmp, f"bundle{i}.bin")
            with open(piece, "wb") as f:
                f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"code{i}.co")
            subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={piece}", f"--output={co}"], check=True)
            out.append(subprocess.run([tools[2], "-d", "--no-show-raw-insn", co], check=True, capture_output=True,
                                      text=True).stdout)
    return out


def _regs(operand):
    """VGPR numbers named by an operand such as v12 or v[70:73]."""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def find_violations(text, slots=2):
    insts, func = [], "?"
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            func = m.group(1)
            continue
        body = line.split("//")[0].strip()
        if not body or body.endswith(":"):
            continue
        parts = body.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        insts.append((func, parts[0], ops, body))
    bad = []
    wide = ("buffer_store_dwordx3", "buffer_store_dwordx4")
    for i, (func, op, ops, body) in enumerate(insts):
        if op not in wide or len(ops) < 4:
            continue
        soffset = ops[3].split()[0]
        if not re.fullmatch(r"s\d+|m0|vcc_lo|vcc_hi|ttmp\d+", soffset):
            continue                                   # immediate scalar offset: the compiler's own wait state applies
        data = _regs(ops[0])
        for f2, op2, ops2, body2 in insts[i + 1:i + 1 + slots]:
            if f2 != func:
                break
            if op2.startswith("v_") and ops2 and not op2.startswith("v_cmp") and (_regs(ops2[0].split()[0]) & data):
                bad.append((func, body, body2))
    return bad


# ---- second guard: the asm MFMAs of the three-term inference kernel ------------------------------------------------
# csrc/mlp_x3i.hip issues v_mfma_f32_16x16x32_bf16 through inline asm (destination = C operand, in place).  The compiler
# does not know what the asm statement is, so it inserts none of the wait states the hardware needs between an MFMA and
# a vector, LDS or memory instruction that reads (or overwrites) its result; the kernel covers the one place where that
# happens -- the end of a layer -- with s_nop.  This check fails the build if any other instruction than an MFMA
# accumulating into the same registers touches an asm MFMA's destination within MFMA_WAIT_STATES wait states.
MFMA_WAIT_STATES = 12      # an 8-pass MFMA's result needs 11 before a vector instruction may read it
ASM_MFMA = "v_mfma_f32_16x16x32_bf16"


def find_mfma_violations(text, need=MFMA_WAIT_STATES):
    insts, func = [], "?"
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            func = m.group(1)
            continue
        body = line.split("//")[0].strip()
        if not body or body.endswith(":"):
            continue
        parts = body.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        insts.append((func, parts[0], ops, body))
    bad = []
    for i, (func, op, ops, body) in enumerate(insts):
        if op != ASM_MFMA or "x3i" not in func:
            continue
        dst = _regs(ops[0])
        states = 0
        for f2, op2, ops2, body2 in insts[i + 1:i + 1 + need + 1]:
            if f2 != func or states >= need:
                break
            if op2 == ASM_MFMA:
                if _regs(ops2[0]) == dst:
                    break                               # the next link of the same accumulation chain: forwarded in hardware
                touched = set().union(*[_regs(o.split()[0]) for o in ops2[1:3]])    # A and B operands
            else:
                touched = set().union(*[_regs(o.split()[0]) for o in ops2]) if ops2 else set()
            if touched & dst:
                bad.append((func, body, body2, states))
                break
            states += 1 + (int(ops2[0]) if op2 == "s_nop" and ops2 and ops2[0].isdigit() else 0)
    return bad


def test_mfma_checker_sees_a_read_too_close():
    sample = """
0000000000001000 <fwd_x3i_k>:
	v_mfma_f32_16x16x32_bf16 v[10:13], v[20:23], v[30:33], v[10:13]
	v_mfma_f32_16x16x32_bf16 v[10:13], v[20:23], v[34:37], v[10:13]
	v_max_f32_e32 v40, 0, v11
	v_mfma_f32_16x16x32_bf16 v[50:53], v[20:23], v[30:33], v[50:53]
	s_nop 15
	v_max_f32_e32 v41, 0, v51
	v_mfma_f32_16x16x32_bf16 v[60:63], v[20:23], v[30:33], v[60:63]
	s_nop 4
	ds_read_b128 v[60:63], v3
"""
    bad = find_mfma_violations(sample)
    assert len(bad) == 2 and "v11" in bad[0][2] and "ds_read_b128" in bad[1][2]


def test_nothing_touches_an_asm_mfma_result_too_early():
    texts = _device_disassembly()
    assert sum(len(re.findall(ASM_MFMA, t)) for t in texts) > 0, "the three-term inference kernel was not found in the disassembly"
    bad = [v for t in texts for v in find_mfma_violations(t)]
    assert not bad, "\n".join(f"{f}: `{a}` then after {n} wait states `{b}`" for f, a, b, n in bad[:10])


def test_checker_sees_the_hazard_pattern():
    sample = """
0000000000001000 <k>:
	buffer_store_dwordx4 v[70:73], v162, s[8:11], s33 offen nt
	v_mov_b32_e32 v71, v5
	buffer_store_dwordx4 v[70:73], v162, s[8:11], 0 offen nt
	v_mov_b32_e32 v71, v5
	buffer_store_dwordx4 v[10:13], v162, s[8:11], s33 offen
	ds_read_b128 v[10:13], v3
	v_add_f32_e32 v14, v1, v2
	buffer_store_dwordx4 v[20:23], v162, s[8:11], s4 offen
	s_nop 0
	v_pk_mul_f32 v[22:23], v[2:3], v[4:5]
"""
    bad = find_violations(sample)
    assert len(bad) == 2 and "s33 offen nt" in bad[0][1] and "v_pk_mul_f32" in bad[1][2]


def test_no_wide_store_with_register_offset_is_followed_by_a_write_of_its_data():
    texts = _device_disassembly()
    stores = sum(len(re.findall(r"buffer_store_dwordx4", t)) for t in texts)
    assert stores > 0, "the fused kernels' buffer stores were not found in the disassembly"
    bad = [v for t in texts for v in find_violations(t)]
    assert not bad, "\n".join(f"{f}: `{a}` then `{b}`" for f, a, b in bad[:10])

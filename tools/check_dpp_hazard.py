#!/usr/bin/env python
"""Static check of the gfx950 code objects for the DPP read-after-VALU-write hazard.

A DPP instruction (v_*_dpp: row_newbcast / row_shr / quad_perm ... on its first source operand) that reads a VGPR written by a
VALU instruction needs TWO wait states between the two.  The hardware does not interlock, the compiler's hazard recognizer
inserts them for the code it generates - but it does not look inside `asm` statements, and k_dpp.h / k_chol2.hip / k_feat.hip
issue their broadcast-in-FMA chains (v_fmac_f64_dpp ... row_newbcast) from inline asm, with the wait states written out next to the
producer (`s_nop 1`).  A compiler-inserted VALU write of the DPP operand between producer and consumer (a v_mov from live-range
splitting, a v_accvgpr_read of a spill reload) would read a stale lane value without any diagnostic; this tool finds that in the
final machine code.  It is run by tests/test_capi_cpu.py on every build.

Method: extract the gfx950 code object of every _obj/*.o (llvm-objdump --offloading), disassemble, and for every DPP instruction
walk back over the preceding instructions until two wait states have passed (every instruction is one wait state, `s_nop N` is
N + 1); any v_* instruction among them whose destination overlaps the DPP source register is a violation.  Straight-line order
only (a label resets the window: the chains in question are branch-free).

Usage: python tools/check_dpp_hazard.py [OBJ ...]        exit code 1 on a violation."""
from __future__ import annotations

import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
_REG = re.compile(r"^(v|a)(?:\[(\d+):(\d+)\]|(\d+))$")
_DPP_MOD = re.compile(r"\b(row_newbcast|row_shl|row_shr|row_ror|row_bcast|row_mirror|row_half_mirror|row_share|row_xmask|quad_perm|wave_shl|wave_shr|wave_rol|wave_ror|dpp8)")


def reg_range(tok):
    """('v', lo, hi) of an operand like v12 or v[12:13]; None for anything else."""
    m = _REG.match(tok.strip().lstrip("-|").rstrip("|"))
    if not m:
        return None
    if m.group(4) is not None:
        k = int(m.group(4))
        return m.group(1), k, k
    return m.group(1), int(m.group(2)), int(m.group(3))


def operands(text):
    """Mnemonic and comma-separated operands (modifiers behind the last operand stay attached to it and are cut at the first blank)."""
    parts = text.split(None, 1)
    if len(parts) == 1:
        return parts[0], []
    ops = [o.strip() for o in parts[1].split(",")]
    ops = [o.split()[0] if o.split() else o for o in ops]
    return parts[0], ops


def check_disassembly(lines, name):
    bad = []
    window = []  # (wait states this instruction accounts for, mnemonic, dest register range or None, text)
    func = "?"
    n_dpp = 0
    for ln in lines:
        if ln.endswith(":") and not ln.startswith("\t"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
            if m:
                func = m.group(1)
            window = []
            continue
        if not ln.startswith("\t"):
            continue
        text = ln.split("//", 1)[0].strip()
        if not text:
            continue
        mnem, ops = operands(text)
        if mnem.startswith("v_") and _DPP_MOD.search(text):
            n_dpp += 1
            # src0 of a DPP instruction: the operand behind the destination (v_cmp*_dpp writes vcc / an SGPR pair: same position)
            src = reg_range(ops[1]) if len(ops) > 1 else None
            if src is not None:
                ws = 0
                for w_states, w_mnem, w_dst, w_text in reversed(window):
                    if ws >= 2:
                        break
                    if w_mnem.startswith("v_") and w_dst is not None and w_dst[0] == src[0] and not (w_dst[2] < src[1] or w_dst[1] > src[2]):
                        bad.append("%s: %s\n      writes the DPP operand %d wait state(s) in front of\n      %s   [%s]" % (name, w_text, ws, text, func))
                        break
                    ws += w_states
        if mnem == "s_nop":
            try:
                states = int(ops[0], 0) + 1
            except (ValueError, IndexError):
                states = 1
            window.append((states, mnem, None, text))
        else:
            dst = reg_range(ops[0]) if (mnem.startswith("v_") and ops and not mnem.startswith("v_cmp")) else None
            window.append((1, mnem, dst, text))
        if len(window) > 8:
            window.pop(0)
    return bad, n_dpp


def check_object(path):
    tmp = tempfile.mkdtemp(prefix="ovp_dpp_")
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [f for f in glob.glob(local + ".*") if "gfx950" in f]
        bad, n = [], 0
        for co in cos:
            out = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
            b, k = check_disassembly(out.splitlines(), os.path.basename(path))
            bad += b
            n += k
        return bad, n, len(cos)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main(argv):
    objs = argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ov_plane_amd", "csrc", "_obj", "*.o")))
    if not objs:
        print("no objects to check (build first)")
        return 2
    total_bad = []
    for o in objs:
        bad, n, ncos = check_object(o)
        print("%-16s %6d DPP instructions in %d gfx950 code object(s), %d hazard(s)" % (os.path.basename(o), n, ncos, len(bad)))
        total_bad += bad
    for b in total_bad:
        print("HAZARD " + b)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))

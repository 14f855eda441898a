#!/usr/bin/env python3
"""Static check of gfx950 assembly for the code-generation defect behind the 'alpha_dual = 1' finding.

The defect: the register allocator splits a VGPR's live range around a call (the value is parked in an AGPR because the callee clobbers
v0..v247) and places the copy at the TOP of a control-flow join block -- in front of the `s_or_b64 exec, exec, s[..]` that restores EXEC
there.  At a loop exit or behind an `s_cbranch_execz` EXEC is empty at that point, so the per-lane copy (v_accvgpr_write_b32 / v_mov_b32,
which obey EXEC; v_writelane_b32 does not) copies NOTHING, and the reload after the call brings back stale bytes.

The check flags every EXEC-obeying vector copy that sits between a basic-block label and the first `s_or_b64 exec, exec, ...` of that block
when nothing but scalar instructions, lane writes / reads and no-ops precedes that restore (i.e. the restore is the block's control-flow
prologue).  A second check (lint_asm_mfma) guards the inline-assembly MFMAs of kkt_fused.hip.  usage: isa_lint.py file.s [...]   or   isa_lint.py --compile src.hip [hipcc flags ...]   (exit code 1 if anything is flagged)"""
import re
import subprocess
import sys
import tempfile

COPY = re.compile(r"^\s*(v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32|v_mov_b32_e32|v_mov_b64_e32|v_mov_b32|v_mov_b64)\b")
HARMLESS = re.compile(r"^\s*((s_\w+|v_writelane_b32|v_readlane_b32|v_readfirstlane_b32)\b.*|;.*)?$")
RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
LABEL = re.compile(r"^(\.LBB[\w.]+|[A-Za-z_][\w.$]*):|^; %bb\.\d+:")   # (fall-through blocks are only marked by a comment)
BRANCH = re.compile(r"^\s*(s_cbranch_\w+|s_branch|s_setpc_b64|s_swappc_b64|s_endpgm)\b")
FUNC = re.compile(r"^([A-Za-z_][\w.$]*):")


def lint_text(text, name="<asm>"):
    """Returns [(file, line number, function, instruction)] of flagged copies."""
    hits, func = [], "?"
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        m = LABEL.match(lines[i])
        if not m:
            i += 1
            continue
        if m.group(1) and not lines[i].startswith(".LBB"):
            func = m.group(1)
        pending = []
        j = i + 1
        while j < len(lines):
            ln = lines[j]
            if LABEL.match(ln) or BRANCH.match(ln) or ln.lstrip().startswith((".Lfunc_end", ".section", ".size")):
                break
            if RESTORE.match(ln):
                for h in pending:
                    hits.append((name, h[0] + 1, func, h[1].strip()))
                break
            if COPY.match(ln):
                pending.append((j, ln))
            elif not HARMLESS.match(ln):
                break            # a real instruction before any restore: whatever restore follows is not this block's prologue
            j += 1
        i += 1
    return hits


ASM_MFMA = re.compile(r"^\s*v_mfma_\w+\s+v\[(\d+):(\d+)\]")
NOP = re.compile(r"^\s*s_nop\s+(\d+)")
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
MFMA_RESULT_WAIT = 18     # 16-pass DGEMM result -> VALU / LDS / memory access (the longest of the cases)


def lint_asm_mfma(text, name="<asm>"):
    """An MFMA written as inline assembly with a VGPR destination is invisible to hipcc's hazard recogniser: nothing but another MFMA may
    touch its destination registers until 18 wait states have passed (kkt_fused.hip, JStream::jtj_mfma).  Flags every other instruction
    that does -- a spill of the tile, a copy the register allocator inserted -- in straight-line order after the MFMA (wait states counted
    from s_nop's and one per instruction; labels and fall-through keep the window open, a taken branch is not followed).
    Returns [(file, line number, function, instruction)]."""
    hits, func = [], "?"
    in_asm = False
    pending = []          # [registers of the destination tile, wait states since]
    for i, ln in enumerate(text.split("\n")):
        m = FUNC.match(ln)
        if m and not ln.startswith(".L"):
            func, pending = m.group(1), []
        st = ln.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not st or st.startswith((";", ".")) or LABEL.match(ln):
            continue
        code = st.split(";")[0]
        mm = ASM_MFMA.match(code)
        if mm and in_asm:
            pending = [p_ for p_ in pending if p_[1] < MFMA_RESULT_WAIT]
            for p_ in pending:
                p_[1] += 1
            pending.append([set(range(int(mm.group(1)), int(mm.group(2)) + 1)), 0])
            continue
        n = NOP.match(code)
        step = int(n.group(1)) + 1 if n else 1
        if not code.startswith("v_mfma") and not n:
            regs = set()
            for r in VREG.finditer(code):
                regs |= set(range(int(r.group(1)), int(r.group(2)) + 1)) if r.group(1) else {int(r.group(3))}
            for tile, ws in pending:
                if ws < MFMA_RESULT_WAIT and regs & tile:
                    hits.append((name, i + 1, func, st))
                    break
        for p_ in pending:
            p_[1] += step
    return hits


def main(argv):
    if len(argv) >= 2 and argv[0] == "--compile":
        out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
        subprocess.run(["/opt/rocm/bin/hipcc", "-S", "--cuda-device-only", "-o", out] + argv[2:] + [argv[1]], check=True,
                       stderr=subprocess.DEVNULL)
        files = [out]
    else:
        files = argv
    bad = 0
    for f in files:
        text = open(f).read()
        hits = lint_text(text, f)
        for h in hits:
            print("%s:%d: [%s] vector copy in front of the EXEC restore of its block: %s" % h)
        bad += len(hits)
        hits = lint_asm_mfma(text, f)
        for h in hits:
            print("%s:%d: [%s] touches the result of an inline-assembly MFMA inside its 18 wait states: %s" % h)
        bad += len(hits)
    print(f"isa_lint: {bad} flagged in {len(files)} file(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

"""Static checks on the EMITTED gfx950 code of libtavhip.so (llvm-objdump of the embedded code objects).  Runs without a GPU; the CPU test
suite calls it (tests/test_abi_and_host.py::test_emitted_isa_discipline) and `__graft_entry__.build()` can be followed by it by hand:

    python tools/check_isa.py [path/to/libtavhip.so]

hipcc neither pads hazards inside inline asm nor knows what an asm statement does to M0, so two hand-maintained invariants of csrc/common.h
are verified on the binary itself.  (Declaring "m0" as a clobber of glds16_m0 is not an alternative: hipcc 7.2 answers `inline asm clobber list
contains reserved registers: m0 ... may lead to undefined behaviour` -- M0 is not allocatable, there is nothing for the compiler to keep out of it.)

 1. M0 discipline (glds16_s / glds16_m0 / glds16_x4).  The GEMM main loops write M0 WITHOUT saving it.  That is only legal while nothing the
    compiler generates uses M0.  Checked: every instruction that names m0 is `s_mov_b32 m0, x`, `s_add_u32 m0, x, y` or `s_mov_b32 sN, m0`;
    every M0 write is followed -- over at most one s_nop -- by a global_load_lds_dwordx4, or is the restore that directly follows one; and no
    other implicit M0 user (ds_gws*, s_sendmsg*, *movrel*, v_interp*, ds_*addtid*, s_set_gpr_idx*, buffer_load ... lds) exists in the library.

 2. to_sgpr (opaque v_readfirstlane_b32 with hand-placed s_nops; round 2 lost a day to this one).  For EVERY v_readfirstlane_b32 sX, vY
    in the library, compiler-made ones included: the instruction in front of it does not write vY (VALU write -> lane read needs one wait
    state), no vector-memory instruction reads sX within the next 5 wait states and no VALU instruction within the next 2 (a VALU-written
    SGPR is not interlocked for those readers on gfx90a+).  s_nop N counts N + 1 wait states, any other instruction one.
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
M0_IMPLICIT = re.compile(r"^(ds_gws|s_sendmsg|s_movrel|v_movrel|v_interp|ds_\w*addtid|s_ttrace|s_set_gpr_idx)")     # (s_set_gpr_idx_on/off/idx: gfx9 VGPR indexing writes and uses M0 without naming it)
INST = re.compile(r"^\s*([a-z_0-9]+)\s*(.*?)\s*(?://.*)?$")


def disassemble(so_path):
    tmp = tempfile.mkdtemp(prefix="tavisa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        objs = sorted(glob.glob(local + ".*gfx950"))
        if not objs:
            raise RuntimeError("no gfx950 code object found in " + so_path)
        out = []
        for o in objs:
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
            out.append((os.path.basename(o), txt))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def sregs(text):
    """SGPR indices named in an operand string (s5, s[4:7])."""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(x) for x in re.findall(r"\bs(\d+)\b", text))
    return regs


def vregs(text):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return regs


def wait_states(op, args):
    if op == "s_nop":
        try:
            return int(args.strip() or "0", 0) + 1
        except ValueError:
            return 1
    return 1


def check(so_path):
    return check_text(disassemble(so_path))


def check_text(objects):
    """objects: [(name, llvm-objdump -d text)] -> (problems, stats)"""
    problems, stats = [], {"m0_writes": 0, "lds_dma": 0, "readfirstlane": 0, "instructions": 0}
    for obj, txt in objects:
        func, insts = "?", []
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                func = m.group(1)
                insts.append(("<label>", "", func))
                continue
            m = INST.match(line)
            if m and m.group(1) and not line.startswith(("Disassembly", obj)) and re.match(r"^\s", line):
                insts.append((m.group(1), m.group(2), func))
        stats["instructions"] += len(insts)
        n = len(insts)
        for i, (op, args, fn) in enumerate(insts):
            if op == "<label>":
                continue
            where = f"{obj}:{fn[:60]}"
            if op.startswith("global_load_lds") or (op.startswith("buffer_load") and re.search(r"\blds\b", args)):
                stats["lds_dma"] += 1
                if op.startswith("buffer_load"):
                    problems.append(f"{where}: unexpected LDS-DMA form `{op} {args}`")
            if M0_IMPLICIT.match(op):
                problems.append(f"{where}: implicit M0 user `{op} {args}`")
            if re.search(r"\bm0\b", args):
                dst = args.split(",")[0].strip()
                ok_write = dst == "m0" and op in ("s_mov_b32", "s_add_u32", "s_add_i32", "s_or_b32")
                ok_read = op == "s_mov_b32" and re.match(r"^s\d+$", dst) and args.split(",")[1].strip() == "m0"
                if not (ok_write or ok_read):
                    problems.append(f"{where}: M0 touched by `{op} {args}`")
                if ok_write:
                    stats["m0_writes"] += 1
                    j = i + 1
                    while j < n and insts[j][0] == "s_nop":
                        j += 1
                    follows_dma = j < n and insts[j][0].startswith("global_load_lds")
                    k = i - 1
                    restores = k >= 0 and insts[k][0].startswith("global_load_lds")
                    if not (follows_dma or restores):
                        problems.append(f"{where}: M0 write `{op} {args}` neither feeds an LDS-DMA nor restores M0 right after one")
                    if follows_dma and j == i + 1:
                        problems.append(f"{where}: no wait state between `{op} {args}` and the LDS-DMA that reads M0")
            if op == "v_readfirstlane_b32":
                stats["readfirstlane"] += 1
                parts = [a.strip() for a in args.split(",")]
                sx, vy = sregs(parts[0]), vregs(parts[1]) if len(parts) > 1 else set()
                if i > 0:
                    pop, pargs, _ = insts[i - 1]
                    if pop.startswith("v_") and not pop.startswith("v_cmp") and pargs:
                        pdst = vregs(pargs.split(",")[0])
                        if pdst & vy:
                            problems.append(f"{where}: `{pop} {pargs}` writes the VGPR that the next instruction, v_readfirstlane_b32 {args}, reads (needs a wait state)")
                ws, j = 0, i + 1
                while j < n and ws < 5:
                    jop, jargs, _ = insts[j]
                    if jop == "<label>":
                        break
                    reads = ",".join(jargs.split(",")[1:]) if jargs else ""
                    if jop.startswith(("global_", "buffer_", "flat_", "scratch_")) and (sregs(jargs) & sx):
                        problems.append(f"{where}: `{jop} {jargs}` reads s{sorted(sx)} {ws} wait states after v_readfirstlane_b32 {args} (needs 5)")
                    if ws < 2 and jop.startswith("v_") and jop != "v_readfirstlane_b32" and (sregs(reads) & sx):
                        problems.append(f"{where}: `{jop} {jargs}` reads s{sorted(sx)} {ws} wait states after v_readfirstlane_b32 {args} (needs 2)")
                    if sregs(jargs.split(",")[0] if jargs else "") & sx and jop.startswith("s_"):
                        break                                   # the SGPR was overwritten by a scalar instruction: later readers see that value
                    ws += wait_states(jop, jargs)
                    j += 1
    return problems, stats


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "multi-modal-emotion_amd", "libtavhip.so")
    problems, stats = check(so)
    print(f"check_isa: {stats['instructions']} instructions, {stats['lds_dma']} LDS-DMA, {stats['m0_writes']} M0 writes, {stats['readfirstlane']} v_readfirstlane_b32")
    for p in problems[:40]:
        print("PROBLEM", p)
    print("problems:", len(problems))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())

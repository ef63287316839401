"""The hazard behind the intermittent memory faults of rounds 3-5 (profiles/HISTORY.md [9]), checked on the BUILT code objects.

gfx9-family hardware needs five wait states between a VALU instruction that writes an SGPR (v_readlane_b32 restoring a spilled scalar,
v_readfirstlane_b32) and a vector-memory instruction that reads that SGPR as its scalar base.  The compiler inserts them for its own instructions;
it cannot see inside inline assembly, where the engine's HBM -> LDS copies (global_load_lds_dwordx4) and write-through stores
(global_store_dwordx4 ... sc1) live.  This script extracts the gfx950 code objects from a library, disassembles them and reports every
vector-memory instruction with a scalar base whose base register was written by a VALU instruction fewer than five wait states earlier
(straight-line scan backwards; `s_nop N` counts N + 1).  Two one-wait-state hazards of the same family are looked for as well (scan_other):
a store of more than 64 bits whose data registers the next instruction overwrites, and an HBM -> LDS copy directly behind a write of M0.

    python scripts/check_asm_hazards.py positionbaseddynamics_amd/_lib/libpbdx.so [more libraries]      exit status 1 if anything is found"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)\w+\s+(.*)$")
SBASE = re.compile(r"s\[(\d+):(\d+)\]")
VALU_SGPR = re.compile(r"^(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+),")


def disassemble(lib):
    """-> {kernel symbol: [instruction text]} of every gfx950 code object bundled in `lib`."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "copy.so")], check=True, capture_output=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
        for k, p in enumerate(starts):
            end = starts[k + 1] if k + 1 < len(starts) else len(data)
            b = os.path.join(tmp, "bundle%d.bin" % k)
            co = os.path.join(tmp, "dev%d.co" % k)
            open(b, "wb").write(data[p:end])
            subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + b, "--output=" + co],
                           check=True, capture_output=True)
            text = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            name = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
                if m:
                    name = m.group(1)
                    out[name] = []
                    continue
                if name is None or not line.startswith("\t"):
                    continue
                out[name].append(line.strip().split("//")[0].strip())
    return out


def scan(instructions, need=5):
    """-> [(index, wait states, memory instruction, VALU instruction)] for every violation in one kernel."""
    found = []
    for i, t in enumerate(instructions):
        m = VMEM.match(t)
        if not m:
            continue
        sb = SBASE.findall(m.group(2))
        if not sb:
            continue
        # the scalar base is the LAST s[a:b] operand of a global / scratch instruction with saddr; buffer resources are s[a:a+3] and never VALU-written here
        lo, hi = int(sb[-1][0]), int(sb[-1][1])
        if hi - lo != 1:
            continue
        ws = 0
        for j in range(i - 1, max(i - 16, -1), -1):
            u = instructions[j]
            mm = VALU_SGPR.match(u)
            if mm and lo <= int(mm.group(2)) <= hi:
                if ws < need:
                    found.append((i, ws, t, u))
                break
            n = re.match(r"^s_nop (\d+)", u)
            ws += int(n.group(1)) + 1 if n else 1
            if ws >= need:
                break
    return found


WIDE_STORE = re.compile(r"^(global_store|flat_store|scratch_store|buffer_store)_dwordx[34]\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")
LDS_DMA = re.compile(r"^(global_load_lds_\w+|buffer_load_\w+\s.*\blds\b)")


def _vregs(operand):
    m = VREG.search(operand)
    if not m:
        return None
    return (int(m.group(1)), int(m.group(2))) if m.group(1) is not None else (int(m.group(3)), int(m.group(3)))


def scan_other(instructions):
    """Two more hazards of the same family that the compiler cannot see through inline assembly (one wait state each):
    (a) a vector-memory store of more than 64 bits followed directly by a VALU write of its data registers;
    (b) an SALU write of M0 followed directly by an HBM -> LDS copy (which takes its LDS base from M0).
    -> [(index, what, instruction, previous / next instruction)]"""
    found = []
    for i, t in enumerate(instructions):
        m = WIDE_STORE.match(t)
        if m and i + 1 < len(instructions):
            ops = [o.strip() for o in m.group(2).split(",")]
            data = _vregs(ops[0] if m.group(1) == "buffer_store" else (ops[1] if len(ops) > 1 else ""))
            nxt = instructions[i + 1]
            if data and nxt.startswith("v_") and not nxt.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
                dst = _vregs(nxt.split(None, 1)[1].split(",")[0]) if " " in nxt else None
                if dst and dst[0] <= data[1] and data[0] <= dst[1]:
                    found.append((i, "store data overwritten in the next cycle", t, nxt))
        if LDS_DMA.match(t) and i > 0 and re.match(r"^s_\w+\s+m0\b", instructions[i - 1]):
            found.append((i, "M0 written by the instruction before", t, instructions[i - 1]))
    return found


def main(argv):
    bad = 0
    for lib in argv:
        kernels = disassemble(lib)
        total = sum(len(v) for v in kernels.values())
        sites = other = 0
        for name, ins in sorted(kernels.items()):
            for i, ws, t, u in scan(ins):
                sites += 1
                if sites <= 8:
                    print("%s: %s: `%s` only %d wait state(s) after `%s`" % (os.path.basename(lib), name[:70], t, ws, u))
            for i, what, t, u in scan_other(ins):
                other += 1
                if other <= 8:
                    print("%s: %s: `%s`: %s (`%s`)" % (os.path.basename(lib), name[:70], t, what, u))
        print("%s: %d kernels, %d instructions, %d hazard site(s), %d of the one-wait-state kinds" % (os.path.basename(lib), len(kernels), total, sites, other))
        bad += sites + other
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

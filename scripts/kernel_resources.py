#!/usr/bin/env python3
"""No GPU: registers, spills, scratch and code size of the gfx950 kernels in a BUILT library (code-object metadata + symbol table), and optionally the ISA
of one kernel.   usage: kernel_resources.py lib.so [name substring] [--isa out.s]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
lib = args[0]; flt = args[1] if len(args) > 1 else ""
isa_out = sys.argv[sys.argv.index("--isa") + 1] if "--isa" in sys.argv else None
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "copy.so")], check=True, capture_output=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    for k, p in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(data)
        b = os.path.join(tmp, "b%d" % k); co = os.path.join(tmp, "d%d.co" % k)
        open(b, "wb").write(data[p:end])
        subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + b, "--output=" + co], check=True, capture_output=True)
        sizes = {}
        for line in subprocess.run([LLVM + "llvm-readelf", "-sW", co], capture_output=True, text=True).stdout.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC":
                sizes[f[7]] = int(f[2])
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        # one block per kernel: "- .agpr_count:" starts it (keys are sorted alphabetically)
        for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            kv = dict(re.findall(r"\.(\w+):\s+(\S+)", ".agpr_count: " + blk))
            sym = kv.get("name", "")
            name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
            if flt not in name:
                continue
            print("%-62s vgpr %3s agpr %2s sgpr %3s | spilled v %2s s %3s | scratch %4s B | lds static %5s B | code %6d B" % (
                name.replace("(anonymous namespace)::", "")[:62], kv.get("vgpr_count"), kv.get("agpr_count"), kv.get("sgpr_count"), kv.get("vgpr_spill_count"),
                kv.get("sgpr_spill_count"), kv.get("private_segment_fixed_size"), kv.get("group_segment_fixed_size"), sizes.get(sym, 0)))
            if isa_out:
                text = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
                m = re.search(r"^[0-9a-f]+ <%s>:\n(.*?)(?=^[0-9a-f]+ <|\Z)" % re.escape(sym), text, re.S | re.M)
                if m:
                    open(isa_out, "w").write(m.group(1)); isa_out = None

"""Instruction histogram of one kernel (or of an address range of it) in the built library's gfx950 code:
python tools/isa_hist.py <kernel name substring> [lo hi]   (addresses in hex, as llvm-objdump prints them)"""
import collections, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("GF_LIB_PATH") or os.path.join(root, "gyroflow_amd", "libgfremap.so")
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
d = tempfile.mkdtemp()
subprocess.check_call(["cp", lib, d + "/lib.so"])
subprocess.check_call([objdump, "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL)
co = [f for f in os.listdir(d) if "amdgcn" in f][0]
txt = subprocess.check_output([objdump, "-d", d + "/" + co]).decode()
name = sys.argv[1]
m = [x for x in re.finditer(r"^[0-9a-f]+ <(\S+)>:$", txt, re.M) if name in x.group(1)]
for x in m:
    body = txt[x.end():txt.find("\n\n", x.end())]
    ins = []
    for l in body.split("\n"):
        mm = re.match(r"\s+(\S+)\s+(.*?)//\s*([0-9A-F]+):", l)
        if mm:
            ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2).strip()))
    if len(sys.argv) > 3:
        lo, hi = int(sys.argv[2], 16), int(sys.argv[3], 16)
        ins = [i for i in ins if lo <= i[0] < hi]
    print(x.group(1), len(ins), "instructions")
    if len(sys.argv) <= 3:
        for a, op, args in ins:
            if op.startswith(("s_cbranch", "s_branch")):
                off = int(args.split()[0])
                if off >= 32768:
                    print("   backward branch %x -> %x" % (a, a + 4 + (off - 65536) * 4))
    c = collections.Counter(op for a, op, args in ins if not op.startswith("s_"))
    print("  ", sorted(c.items(), key=lambda kv: -kv[1])[:40])
    print("   salu+branch:", sum(1 for a, op, args in ins if op.startswith("s_")))
    if os.environ.get("ISA_DUMP"):
        open(os.environ["ISA_DUMP"], "w").write("\n".join("%x %s %s" % i for i in ins))

"""Static audit of the library's gfx950 code for two patterns that cost memory round trips inside loops:
a load followed (within three instructions) by s_waitcnt vmcnt(0), and loads behind per-element branches.
Compiles singa_hip.hip to assembly and prints, per kernel: such load->wait pairs, all vmcnt(0) waits, loads, branches.
    python tools/isa_audit.py [top N]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "singa_audit.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w", "-o", out,
                       os.path.join(ROOT, "singa_amd", "csrc", "singa_hip.hip")])
kern, cur = {}, None
for ln in open(out):
    m = re.match(r"^(_Z[\w]+):", ln)
    if m:
        cur = m.group(1)
        kern[cur] = []
        continue
    if ln.strip().startswith(".amdhsa_kernel"):
        cur = None
    if cur:
        kern[cur].append(ln.strip())
names = subprocess.run(["c++filt"], input="\n".join(kern), capture_output=True, text=True).stdout.split("\n")
LOAD = ("global_load", "buffer_load", "flat_load")
rows = []
for (k, body), name in zip(kern.items(), names):
    ins = [b for b in body if b and not b.startswith((";", "."))]
    loads = sum(b.startswith(LOAD) for b in ins)
    w0 = sum(b.startswith("s_waitcnt") and "vmcnt(0)" in b for b in ins)
    br = sum(b.startswith("s_cbranch") for b in ins)
    ser = 0
    for i, b in enumerate(ins):
        if b.startswith(LOAD):
            for j in range(i + 1, min(i + 4, len(ins))):
                if ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]:
                    ser += 1
                    break
                if ins[j].startswith(LOAD):
                    break
    rows.append((ser, w0, loads, br, name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")))
rows.sort(reverse=True)
print("load->vmcnt(0)  vmcnt(0)  loads  branches  kernel")
for r in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 40]:
    print("%10d %10d %6d %9d   %s" % r)

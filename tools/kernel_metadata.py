"""Register / spill / scratch metadata of every kernel, from `hipcc -S` of each device translation unit with the library's
flags (no GPU needed).  Writes profiles/<name> (default round3_kernel_metadata.txt)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fugue_amd import build as B
out = os.path.join(ROOT, "profiles", os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "round4_kernel_metadata.txt")   # never outside profiles/
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-DFG_BUILD", "-Wno-unused-function", "-w"]
rows = []
for src in [s for s in B.SOURCES if s.endswith(".hip")]:
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run([B.hipcc()] + flags + ["-x", "hip", os.path.join(B.CSRC, src), "--cuda-device-only", "-S", "-o", asm], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t = open(asm).read()
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", t, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(k + r":\s+(\d+)", body).group(1))
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows.append((src, name, g(r"\.vgpr_count"), g(r"\.sgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.sgpr_spill_count"), g(r"\.private_segment_fixed_size")))
with open(out, "w") as f:
    f.write("# hipcc -S metadata (gfx950, the library's flags): VGPRs, SGPRs, spilled VGPRs, spilled SGPRs (to VGPR lanes), scratch bytes per lane\n")
    f.write("%-16s %-58s %5s %5s %7s %7s %8s\n" % ("file", "kernel", "vgpr", "sgpr", "v-spill", "s-spill", "scratch"))
    for r in rows:
        f.write("%-16s %-58s %5d %5d %7d %7d %8d\n" % r)
print(open(out).read())

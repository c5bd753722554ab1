"""Static instruction mix of the bundle kernel k_fim_bundle<G, threads, members per lane> (default <16, 256, 2>; gfx950, the tree's flags): whole kernel and the loop nest around the solver
(member loop inside node-trip loop inside half-round loop inside round loop), from the compiler's own assembly.
   python3 tools/isa_bundle.py [G [threads [members per lane]]] > profiles/r04_isa_bundle_kernel.txt      (CPU only: hipcc cross-compiles)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsurftomo_amd import build
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 256
MPL = int(sys.argv[3]) if len(sys.argv) > 3 else 2
with tempfile.TemporaryDirectory() as td:
    cmd = [build.hipcc()] + build.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-save-temps", "-c", os.path.join(ROOT, "dsurftomo_amd", "csrc", "bundle_kernel.hip"), "-o", os.path.join(td, "b.o")]
    r = subprocess.run(cmd, cwd=td, capture_output=True, text=True)
    asm = open([os.path.join(td, f) for f in os.listdir(td) if f.endswith("gfx950.s")][0]).read()
    remarks = r.stderr
name = re.search(r"_ZN3dsa12k_fim_bundleILi%dELi%dELi%dELb%dEEEv\w+" % (G, NT, MPL, int(os.environ.get("DSA_ISA_TIE", "0"))), asm).group(0)
print("k_fim_bundle<%d, %d, %d>: %s" % (G, NT, MPL, " ".join(cmd[1:-5])))
blk = remarks[remarks.index("Function Name: " + name):]
for key in ("VGPRs:", "AGPRs:", "SGPRs:", "ScratchSize", "Occupancy", "SGPRs Spill", "VGPRs Spill", "LDS Size"):
    m = re.search(r"\s(%s[^\n]*?)\s\[-Rpass" % re.escape(key), blk)
    if m: print("   ", m.group(1).strip())
a = asm.index(name + ":"); b = asm.index(".Lfunc_end", a)
lines = asm[a:b].split("\n")


def mix(lo, hi):
    ins = [l.strip() for l in lines[lo:hi + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    v = [x for x in ins if x.startswith("v_")]
    fp = [x for x in v if re.match(r"v_(add|sub|mul|fma|fmac|mac|mad|div|rcp|sqrt|min|max|rsq)\w*_f32", x)]
    return dict(all=len(ins), valu=len(v), fp32=len(fp), mov=sum(x.startswith("v_mov") for x in v), cndmask=sum("cndmask" in x for x in v), cmp=sum(x.startswith("v_cmp") for x in v),
                lanes=sum(("readlane" in x or "writelane" in x or "dpp" in x) for x in v), salu=sum(x.startswith("s_") for x in ins), vmem=sum(x.startswith(("global_", "buffer_", "flat_", "scratch_")) for x in ins),
                lds=sum(x.startswith("ds_") for x in ins), branch=sum(x.startswith(("s_cbranch", "s_branch")) for x in ins), barrier=sum(x.startswith("s_barrier") for x in ins))


def show(tag, d):
    print("%-90s %5d instructions: VALU %4d (fp32 arithmetic %3d, v_mov %3d, v_cndmask %3d, v_cmp %3d, lane ops %3d)  SALU %4d  vector memory %3d  LDS %3d  branches %3d  barriers %d" %
          (tag, d["all"], d["valu"], d["fp32"], d["mov"], d["cndmask"], d["cmp"], d["lanes"], d["salu"], d["vmem"], d["lds"], d["branch"], d["barrier"]))


show("whole kernel", mix(0, len(lines) - 1))
lab = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < i: loops.append((lab[m.group(1)], i))
sq = [i for i, l in enumerate(lines) if "v_sqrt_f32" in l]
# the solver sits where the square roots cluster (pass B); loops around that cluster, innermost first, one per distinct header region
core = sq[len(sq) // 2]
around = sorted([(hi - lo, lo, hi) for lo, hi in loops if lo <= core <= hi])
names = ["member loop (one solve_node per trip; the body exists once)", "node-trip loop (16 nodes x 4 lanes x 4 members per wave trip)", "half-round loop (even nodes, then odd)", "round loop (pass A + pass B + bookkeeping)"]
picked, last = [], -1
for size, lo, hi in around:
    if last < 0 or size > 1.15 * last: picked.append((lo, hi)); last = size
if picked and sum(picked[0][0] <= i <= picked[0][1] for i in sq) > 3:      # the default build: the member body four times inside the node-trip loop, no member loop
    names = ["node-trip loop (16 nodes x 4 lanes per wave trip; four member evaluations, unrolled)"] + names[2:]
for nm, (lo, hi) in zip(names, picked): show(nm, mix(lo, hi))
print("(static counts; rare paths -- exception-table probes, the inactive-member path -- are inside them.  Dynamic, per solve at 1025^2, profiles/pmc_latest.json:")
print(" VALU / SALU wave instructions, fabric bytes, valu_issue.)")

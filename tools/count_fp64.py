"""FP64 vector instructions per thread of the apply kernels, counted in the gfx950 ISA (hipcc -S of the kernel sources with the build's flags):
the `fp64_instructions_per_cell` constants of bench.py's `roofline.fp64_issue`.  One thread = one cell in every apply kernel (the loader waves of
the ring kernels execute no FP64 instruction), and the facet bodies are fully unrolled, so the static count of the kernel IS the count per cell;
a kernel with a back edge around FP64 code is flagged: in the persistent ring kernels that loop is the walk over the blocks (one cell per lane and
trip, so the count is still per cell), anywhere else the static count would be a lower bound.
usage: python tools/count_fp64.py [file.hip ...]        (default: the four apply sources)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "knp-emi-dg_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        return name


def count(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "--cuda-device-only", "-S",
                        os.path.join(CSRC, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for i in starts:
        name = lines[i].split(":")[0]
        j = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        label_at, n64, ninst, back = {}, 0, 0, False
        pos_f64 = []
        for k in range(i, j):
            t = lines[k].strip()
            m = re.match(r"^(\.LBB\d+_\d+):", lines[k])
            if m:
                label_at[m.group(1)] = k
            if re.match(r"v_\w+_f64", t):
                n64 += 1
                pos_f64.append(k)
            if t and not t.startswith((";", ".")) and not t.endswith(":"):
                ninst += 1
        for k in range(i, j):                                       # a branch to an earlier label with FP64 code in between = a loop over it
            m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[k])
            if m and m.group(1) in label_at and label_at[m.group(1)] < k and any(label_at[m.group(1)] < p < k for p in pos_f64):
                back = True
        dem = demangle(name).replace("(anonymous namespace)::", "")
        dem = dem.split("(")[0].replace("void ", "")
        if "apply" in dem:
            print("%-46s %5d FP64  %5d instructions%s" % (dem, n64, ninst, "   (inside a loop: the count is per trip -- the block walk of the persistent kernels, one cell per lane and trip)" if back else ""))


for f in (sys.argv[1:] or ["apply_ring.hip", "apply_ring_u.hip", "apply_p2.hip", "apply_p1.hip"]):
    print("# " + f)
    count(f)

"""Writes the HBM-traffic records of profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) summarised by
tools/pmc_summary.py, stamping each record with the sha256 of the kernel's source file: bench.py reports `roofline.traffic` only when
that hash still matches the tree (a kernel edited after its counter pass reports null).
usage: pmc_traffic_update.py FETCH_SUMMARY.txt WRITE_SUMMARY.txt CELLS [KEY_SUFFIX]     (run in the repo, after the GPU passes)
FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request of wide coalesced reads); WRITE_SIZE is exact."""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel name as pmc_summary.py prints it -> (key in pmc_traffic.json / bench.py, source file)
KERNELS = {
    "k_emi_apply_ring": ("k_emi_apply_ring", "apply_ring.hip"),
    "k_knp_apply_ring<2, 1>": ("k_knp_apply_ring<2>", "apply_ring.hip"),
    "k_emi_apply_p2<3, 256, true>": ("k_emi_apply_p2<3,256,true>", "apply_p2.hip"),
    "k_knp_apply_p2<3, 256, true>": ("k_knp_apply_p2<3,256,true>", "apply_p2.hip"),
    "k_emi_apply_p2<3, 256, false>": ("k_emi_apply_p2<3,256,false>", "apply_p2.hip"),
    "k_knp_apply_p2<3, 256, false>": ("k_knp_apply_p2<3,256,false>", "apply_p2.hip"),
    "k_emi_apply<3>": ("k_emi_apply<3,3>", "apply_p1.hip"),
    "k_knp_apply<3, 2>": ("k_knp_apply<3,2>", "apply_p1.hip"),
    "k_emi_apply_ring_u": ("k_emi_apply_ring_u", "apply_ring_u.hip"),
    "k_knp_apply_ring_u<2>": ("k_knp_apply_ring_u<2>", "apply_ring_u.hip"),
}


def read(path):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
            if m and name:
                out.setdefault(name, {})[m.group(1)] = float(m.group(2))
    return out


fetch, write, cells = read(sys.argv[1]), read(sys.argv[2]), int(sys.argv[3])
suffix = sys.argv[4] if len(sys.argv) > 4 else ""
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
pmc = json.load(open(path))
for raw, (key, src) in KERNELS.items():
    f = next((v["FETCH_SIZE"] for k, v in fetch.items() if k == raw and "FETCH_SIZE" in v), None)
    w = next((v["WRITE_SIZE"] for k, v in write.items() if k == raw and "WRITE_SIZE" in v), None)
    if f is None or w is None:
        continue
    with open(os.path.join(ROOT, "knp-emi-dg_amd", "csrc", src), "rb") as fh:
        sha = hashlib.sha256(fh.read()).hexdigest()[:16]
    pmc[key + suffix] = {"cells_per_launch": cells, "FETCH_SIZE_KiB": round(f), "WRITE_SIZE_KiB": round(w),
                         "traffic_bytes": int(round((2.0 * f + w) * 1024)), "source": src, "source_sha16": sha}
    print(key + suffix, pmc[key + suffix])
json.dump(pmc, open(path, "w"), indent=1)

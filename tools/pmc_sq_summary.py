"""Per-kernel SQ counters of one program from rocprofv3 --pmc passes (tools/pmc_run.sh layout: gpurun_out/pmc_<tag>_<i>): sums over all launches of a kernel
and the ratios that describe it -- matrix-pipe busy share, VALU share of the issued instructions, LDS bank-conflict share.
    python tools/pmc_sq_summary.py gpurun_out/pmc_<tag>_1 gpurun_out/pmc_<tag>_2 ..."""
import csv, glob, os, re, sys, collections


def short(name):
    m = re.search(r"gsw_mm_kernelIDF16(b?)_Li(\d)ELb([01])ELi(\d)ELb([01])E", name)
    if m:
        return f"gsw_mm_kernel<EPI {m.group(2)}, {'12' if m.group(3) == '1' else '8'} waves, MT {m.group(4)}{', LN fold' if m.group(5) == '1' else ''}>"
    m = re.search(r"gsw_attn_fwd_kernelIDF16_Li(\d)ELi(\d+)ELb([01])ELb([01])E", name)
    if m:
        return f"gsw_attn_fwd_kernel<QB {m.group(1)}, head_dim {int(m.group(2)) * 8}{', ragged' if m.group(4) == '1' else ''}>"
    m = re.search(r"(gsw_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:50]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[(k, r["Counter_Name"])] += 1
        for (k, c), n in seen.items():
            calls[k] = max(calls[k], n)
rows = []
for k, c in acc.items():
    g = c.get("GRBM_GUI_ACTIVE", 0.0)
    rows.append((c.get("SQ_WAVE_CYCLES", 0.0), k, c, g))
print(f"{'kernel':58s} {'launches':>8s} {'MFMA busy / GPU active':>23s} {'VALU / all insts':>17s} {'MFMA insts / VALU insts':>24s} {'LDS conflict / LDS active':>26s} {'wait-any / wave cycles':>23s}")
for _, k, c, g in sorted(rows, reverse=True)[:16]:
    def ratio(a, b):
        return f"{c[a] / c[b]:.3f}" if c.get(b) else "-"
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs' matrix pipes, GRBM_GUI_ACTIVE over the 8 XCDs: busy share = busy / (128 x active)
    mf = "%.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 128.0)) if g and c.get("SQ_VALU_MFMA_BUSY_CYCLES") else "-"
    insts = c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0) + c.get("SQ_INSTS_VMEM_RD", 0)
    vs = "%.3f" % (c["SQ_INSTS_VALU"] / insts) if insts and c.get("SQ_INSTS_VALU") else "-"
    print(f"{k:58s} {calls[k]:8d} {mf:>23s} {vs:>17s} {ratio('SQ_INSTS_MFMA', 'SQ_INSTS_VALU'):>24s} {ratio('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'):>26s} {ratio('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'):>23s}")

"""Timeline of ONE graph replay of the UNet forward out of a rocprofv3 --kernel-trace csv: per launch (in start order) the kernel, its duration
and the idle gap in front of it, then totals per kernel family (busy time, gap time charged to the kernel that follows the gap).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/small_rows_profile.py 1
    python3 tools/graph_timeline.py DIR [replay index from the end, default 2] [--full]
"""
import csv
import glob
import re
import sys


def short(name: str) -> str:
    m = re.search(r"gsw_mm_kernelIDF16_Li(\d)ELb(\d)ELi(\d)ELb(\d)", name)
    if m:
        return f"mm<EPI{m.group(1)},{'12w' if m.group(2) == '1' else '8w'},MT{m.group(3)}{',ln' if m.group(4) == '1' else ''}>"
    m = re.search(r"gsw_mm_kernelIDF16_Li(\d)ELb(\d)ELi(\d)", name)
    if m:
        return f"mm<EPI{m.group(1)},{'12w' if m.group(2) == '1' else '8w'},MT{m.group(3)}>"
    m = re.search(r"(gsw_[a-z0-9_]+)", name)
    if m:
        return m.group(1)
    m = re.search(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)", name)
    if m:
        inner = re.search(r"([a-z0-9_]+_kernel_cuda|[A-Za-z]+Functor[A-Za-z_]*|silu_kernel|CatArrayBatchedCopy[a-z_]*)", name)
        return "torch:" + (inner.group(1) if inner else m.group(1))
    return name[:40]


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
    full = "--full" in sys.argv
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a replay starts at every launch of the first kernel that follows the per-step input copies: split at the largest gaps instead -- the
    # host-side loop leaves > 30 us between two replays, launches inside a graph are closer
    starts = [0]
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 30000:
            starts.append(i)
    starts.append(len(rows))
    segs = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
    segs = [s for s in segs if s[1] - s[0] > 100]
    a, b = segs[-back]
    seg = rows[a:b]
    t0 = seg[0][0]
    fam = {}
    prev_end = seg[0][0]
    busy = gap_tot = 0
    for s, e, n in seg:
        k = short(n)
        gap = max(0, s - prev_end)
        f = fam.setdefault(k, [0, 0, 0])
        f[0] += 1
        f[1] += e - s
        f[2] += gap
        busy += e - s
        gap_tot += gap
        if full:
            print(f"{(s - t0) / 1e3:9.1f} us  gap {gap / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {k}")
        prev_end = max(prev_end, e)
    span = seg[-1][1] - t0
    print(f"replay: {len(seg)} launches, span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, gaps {gap_tot / 1e3:.1f} us ({len(segs)} replays seen)")
    print(f"{'kernel':44s} {'calls':>5s} {'busy us':>9s} {'avg':>7s} {'gap us':>8s} {'share':>6s}")
    for k, (c, t, g) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f"{k:44s} {c:5d} {t / 1e3:9.1f} {t / c / 1e3:7.2f} {g / 1e3:8.1f} {(t + g) / span * 100:5.1f}%")


if __name__ == "__main__":
    main()

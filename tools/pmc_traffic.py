"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs of the same command).
    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > traffic.json
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KB; on gfx950 FETCH_SIZE reports HALF of the bytes
of wide coalesced reads (128-byte requests tallied at 64 B), so it is doubled; WRITE_SIZE is taken as is (it equals the algorithmic bytes of
the store-only embed kernel)."""
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            e = out.setdefault(name, [0, 0.0])
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return out


def short(name):
    m = re.search(r"gsw_mm_kernelIDF16(b?)_Li(\d)ELb([01])ELi(\d)E", name)
    if m:
        tile = {"8": "256x320 tile", "4": "256x160 tile", "2": "128x160 tile"}.get(m.group(4), "MT " + m.group(4))
        return f"gsw_mm_kernel<{'bf16' if m.group(1) else 'f16'}, EPI {m.group(2)}, {'12 waves' if m.group(3) == '1' else '8 waves'}, {tile}>"
    m = re.search(r"(gsw_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:60]


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    rows = {}
    for name in set(fetch) | set(write):
        s = short(name)
        e = rows.setdefault(s, {"launches": 0, "FETCH_SIZE_KB_raw": 0.0, "WRITE_SIZE_KB": 0.0})
        if name in fetch:
            e["launches"] += fetch[name][0]
            e["FETCH_SIZE_KB_raw"] += fetch[name][1]
        if name in write:
            e["WRITE_SIZE_KB"] += write[name][1]
    for e in rows.values():
        e["traffic_bytes_total"] = int((2.0 * e["FETCH_SIZE_KB_raw"] + e["WRITE_SIZE_KB"]) * 1024)
        e["traffic_bytes_per_launch"] = e["traffic_bytes_total"] // max(1, e["launches"])
    json.dump({k: rows[k] for k in sorted(rows, key=lambda k: -rows[k]["traffic_bytes_total"]) if k.startswith("gsw_")}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

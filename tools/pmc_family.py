"""bench_e2e.py's `roofline.traffic` source: HBM bytes per launch of the whole matmul-engine family (every gsw_mm_kernel instantiation + the split-K reduce) from
the per-kernel traffic tables of the UNet forward at the two row counts of a step (tools/pmc_traffic.py outputs).
    python3 tools/pmc_family.py <traffic_b128.json> <traffic_b64.json> <batch> > profiles/r05_e2e_dominant_kernel_pmc.json"""
import json
import sys

t128, t64 = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
tot_b, tot_l, per = 0, 0, {}
for tag, t in (("128", t128), ("64", t64)):
    for k, v in t.items():
        if k.startswith("gsw_mm_kernel"):
            tot_b += v["traffic_bytes_total"]
            tot_l += v["launches"]
            per.setdefault(k, {})[tag] = v["traffic_bytes_per_launch"]
json.dump({"kernel": "gsw_mm_kernel", "config": {"batch": batch, "unet": "sd21", "height": 512, "width": 512},
           "traffic_bytes_per_launch": tot_b // max(1, tot_l), "launches_measured": tot_l,
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/r06_profile.sh) of tools/unet_forward_bench.py at 128 rows (the CFG sampling half "
                  "of a step) and 64 rows (the inversion half); (2 x FETCH_SIZE + WRITE_SIZE) KB as MI355X_MICROARCH.md prescribes for gfx950, summed over EVERY instantiation of "
                  "the matmul engine (the family bench.py's `roofline` reports) / their launches; per-kernel tables next to this file.  The doubling is confirmed for the "
                  "engine's 128-byte LDS-DMA requests by a known change of 335.5 MB that the raw counter shows as 158 MB (profiles/r04_pmc_panel_fetch.txt)",
           "per_epilogue_kind": per}, sys.stdout, indent=1)

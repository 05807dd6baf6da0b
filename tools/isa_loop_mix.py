"""Instruction mix of the hottest loop of a kernel, from hipcc's assembly (hipcc -S --cuda-device-only): the innermost loop with the most MFMAs is taken as
the main loop; instructions are counted by issue class.  Evidence for "issue-bound" statements (attention).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only a-watermark-for-diffusion-models_amd/csrc/gswm_attn.hip -o /tmp/attn.s
    python3 tools/isa_loop_mix.py /tmp/attn.s gsw_attn_fwd_kernelIDF16_Li2ELi8ELb1ELb0E
"""
import re
import sys
from collections import Counter


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return "MFMA"
    if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f16", "v_rcp_f16"):
        return "VALU transcendental (quarter rate)"
    if op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")) or "dpp" in op:
        return "VALU cross-lane"
    if op.startswith("v_pk_"):
        return "VALU packed"
    if op.startswith("v_cvt"):
        return "VALU convert"
    if op.startswith(("v_accvgpr",)):
        return "VALU accvgpr moves"
    if op.startswith("v_"):
        return "VALU other"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "SALU / branch"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if key in l.split(":")[0] and ":" in l and not l.startswith(("\t", ".", " ")))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^\ts_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\ts_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            loops.append((label_at[m.group(1)], i))
    def ops(a, b):
        out = []
        for l in body[a:b + 1]:
            m = re.match(r"^\t([a-z_0-9]+)", l)
            if m and not l.startswith("\t."):
                out.append(m.group(1))
        return out
    marker = sys.argv[sys.argv.index("--skip-marker") + 1] if "--skip-marker" in sys.argv else None
    best = max(loops, key=lambda ab: (sum(o.startswith("v_mfma") for o in ops(*ab)) / (1 + sum(1 for (c, d) in loops if ab[0] <= c and d <= ab[1] and (c, d) != ab)), ab[1] - ab[0]))
    o = ops(*best)
    skipped = 0
    if marker:           # the steady-state path: blocks a forward branch jumps over are left out when they carry the marker comment (e.g. "masked tile")
        keep = [True] * len(body)
        for i in range(best[0], best[1] + 1):
            m = re.match(r"^\ts_cbranch_\w+\s+(\.LBB\d+_\d+)", body[i])
            if m and m.group(1) in label_at and i < label_at[m.group(1)] <= best[1] and any(marker in l for l in body[i:label_at[m.group(1)]]):
                for j in range(i + 1, label_at[m.group(1)]):
                    keep[j] = False
        o = []
        for j in range(best[0], best[1] + 1):
            m = re.match(r"^\t([a-z_0-9]+)", body[j])
            if m and not body[j].startswith("\t."):
                if keep[j]:
                    o.append(m.group(1))
                else:
                    skipped += 1
    cnt = Counter(classify(x) for x in o)
    print(f"kernel {key}: main loop = lines {best[0]}..{best[1]} of the kernel body, {len(o)} instructions per iteration on the steady-state path"
          + (f" ({skipped} more in blocks behind a branch, marker '{marker}')" if marker else ""))
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print(f"  {k:36s} {v:5d}  {v / len(o) * 100:5.1f} %")
    top = Counter(x for x in o if classify(x).startswith("VALU")).most_common(12)
    print("  most frequent VALU opcodes: " + ", ".join(f"{k} x{v}" for k, v in top))


if __name__ == "__main__":
    main()

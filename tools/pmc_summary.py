"""Condense rocprofv3 output directories into the JSON summary kept under profiles/.

  python tools/pmc_summary.py --stats DIR --fetch DIR --write DIR --out profiles/rNN_pmc_traffic.json

DIR are rocprofv3 `-d` directories: --stats from `--kernel-trace --stats`, --fetch from a `--pmc FETCH_SIZE`
pass and --write from a separate `--pmc WRITE_SIZE` pass (the two counters do not fit one pass,
MI355X_MICROARCH.md "HBM" / counter table).  Per (kernel, grid) it reports launches, average duration and the
average HBM-side bytes per launch: FETCH_SIZE is in KB and is DOUBLED (gfx950 counts 128-B requests as 64 B for
16-B-per-lane streaming reads, which is how every hot kernel here loads); WRITE_SIZE is in KB, uncorrected.
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def _find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits


def short(name):
    import re
    m = re.search(r"(gemm_nt256p_kernel|gemm_nt256_kernel|gemm_tn256_kernel)<[^>]*>", name)
    if m:
        return m.group(0)
    for tag in ("gemm_nt_kernel", "gemm_tn_kernel", "head_out_bwd", "head_out_fwd", "head_out_finish", "tn_reduce", "attn_bwd_dkv", "attn_bwd_dq",
                "attn_fwd", "bilinear_bwd", "bilinear_fwd", "ln_bwd_kernel", "ln_bwd_reduce", "ln_fwd", "adam_kernel",
                "linear_head", "permute4", "segsum", "cast_kernel", "pixel_shuffle", "zero_stuff2", "patchify", "loss"):
        if tag in name:
            return tag
    return name[:60]


# The persistent GEMM kernels launch one workgroup per CU whatever the problem, so the grid no longer tells shapes
# apart: dispatches of one (kernel, grid) are split into a "large" class (duration >= half of the group's longest
# dispatch) and a "small" class.  The large class of gemm_nt256p_kernel<1, 0> is the heads' 512->512 3x3 conv.
def _rows(d, suffix):
    for f in _find(d, suffix):
        for r in csv.DictReader(open(f)):
            grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            yield (short(r["Kernel_Name"]), grid), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6, r


def _classes(d, suffix, only=None):
    rows = [x for x in _rows(d, suffix) if only is None or x[2].get("Counter_Name") == only]
    longest = defaultdict(float)
    for k, ms, _ in rows:
        longest[k] = max(longest[k], ms)
    for k, ms, r in rows:
        yield k + ("large" if ms >= 0.5 * longest[k] else "small",), ms, r


def durations(d):
    acc = defaultdict(lambda: [0, 0.0])
    for k, ms, _ in _classes(d, "kernel_trace.csv"):
        acc[k][0] += 1
        acc[k][1] += ms
    return acc


def counters(d, name):
    acc = defaultdict(lambda: [0, 0.0])
    for k, _, r in _classes(d, "counter_collection.csv", only=name):
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--mfma", default=None, help="directory of a --pmc MfmaUtil pass (optional)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    dur = durations(a.stats)
    fe = counters(a.fetch, "FETCH_SIZE")
    wr = counters(a.write, "WRITE_SIZE")
    mf = counters(a.mfma, "MfmaUtil") if a.mfma else {}
    rows = []
    for k, (n, ms) in dur.items():
        row = {"kernel": k[0], "grid_threads": k[1], "class": k[2], "launches": n, "avg_ms": ms / n, "total_ms": ms}
        if k in fe and fe[k][0]:
            row["fetch_bytes_per_launch"] = 2.0 * 1024.0 * fe[k][1] / fe[k][0]
        if k in wr and wr[k][0]:
            row["write_bytes_per_launch"] = 1024.0 * wr[k][1] / wr[k][0]
        if k in mf and mf[k][0]:
            row["mfma_util_percent"] = mf[k][1] / mf[k][0]   # rocprofv3 MfmaUtil = MFMA busy cycles / (GPU active cycles x SIMDs)
        if "fetch_bytes_per_launch" in row and "write_bytes_per_launch" in row:
            row["hbm_bytes_per_launch"] = row["fetch_bytes_per_launch"] + row["write_bytes_per_launch"]
        rows.append(row)
    rows.sort(key=lambda r: -r["total_ms"])
    out = {"command": a.command,
           "corrections": "FETCH_SIZE[KB] x 1024 x 2 (gfx950 half-count of 16-B/lane streaming reads); WRITE_SIZE[KB] x 1024",
           "kernels": rows[:40]}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    for r in rows[:12]:
        print(json.dumps(r))


if __name__ == "__main__":
    main()

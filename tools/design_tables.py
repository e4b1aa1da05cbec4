"""DESIGN.md section 5 as tables, from profiles/r06_* ONLY (round-5 review, item 8): python tools/design_tables.py > profiles/r06_tables.md
Inputs: r06_bench_lines.jsonl (tools/r06_lines.sh: every workload's line on one box), r06_pmc_traffic*.json (tools/profile_round.sh:
kernel stats + separate FETCH_SIZE / WRITE_SIZE / MfmaUtil passes), r06_attention_ceiling.jsonl (tools/probe/attn_ceiling.py),
r06_adam_pack_ab.txt.  Missing inputs are skipped with a note."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(ROOT, "profiles", n)


def lines():
    f = P("r06_bench_lines.jsonl")
    if not os.path.exists(f):
        print("*(r06_bench_lines.jsonl missing)*\n")
        return
    print("### Bench lines (one box; `profiles/r06_bench_lines.jsonl`)\n")
    print("| workload | dtype | value | ms / step | dominant kernel: avg ms, achieved, fraction of its peak | replayed from graphs | legs reported beside the headline |")
    print("|---|---|---|---|---|---|---|")
    for l in open(f):
        d = json.loads(l)
        rf = d["roofline"]
        alts = []
        for k, v in d.items():
            if k.startswith("alt_") and isinstance(v, dict) and "value" in v:
                extra = ""
                for kk in ("peak_index_differs_from_headline", "peak_index_differs_from_6term", "peak_index_differs_from_fp32_mode"):
                    if kk in v:
                        extra = f" ({v[kk]} of 1225 peak indices differ)"
                alts.append(f"`{k[4:]}` {v['value']:.2f}{extra}")
        hg = d.get("hip_graph", {})
        unit = "images/s"
        extra_v = ""
        if "sweep_precision" in d and d["sweep_precision"].get("mode") == "certified":
            extra_v = f" (re-run share {100 * d['sweep_precision']['rerun_fraction']:.1f} %)"
        if "cpu_baseline" in d:
            cb = d["cpu_baseline"]
            alts.append(f"CPU oracle {cb['value']:.4g} {unit} on {cb['cores']} threads")
            if "peak_check" in cb:
                pc = cb["peak_check"]
                alts.append(f"CPU re-derivation of {pc['proposals_checked']} proposals: {pc['peak_index_mismatches']} peak-index mismatches, {pc['certified']} certified")
        name = d["config"]["workload"] + ("" if hg.get("mode", "auto") == "auto" else f" (--graphs {hg.get('mode')})")
        # configurations the reference does not have are marked where their numbers stand (bench.py::REFERENCE_STATUS; lines taken before
        # the key existed are marked from the backbone name)
        ext = {"dpt_large14": " -- **extension: the reference has no patch-14 backbone (patch 16 is hard-coded, `vit.py:262,339`); the semantics are this build's, no reference result exists**",
               "dpt_small": " -- **extension: no ViT-S/16 in the reference (DPT-small convention)**"}.get(d["config"].get("backbone"), "")
        name += ext
        print(f"| {name} | {d['dtype']} | **{d['value']:.2f}** {unit}{extra_v} | {d['ms_per_step']:.2f} | {rf['avg_launch_ms']:.2f} ms, {rf['achieved']:.0f} {rf['unit']}, "
              f"**{rf['frac']:.3f}** of {rf['peak']:.0f} | {'yes: ' + hg.get('form', '') if hg.get('replayed') else 'no'} | {'; '.join(alts) or '-'} |")
    print()


def breakdown(fname, steps, title):
    f = P(fname)
    if not os.path.exists(f):
        print(f"*({fname} missing)*\n")
        return
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    d = json.load(open(f))
    rows = d["kernels"]
    src = open(os.path.join(ROOT, "tools", "step_breakdown.py")).read()
    ns = {}
    exec(src[src.index("def grp(k):"):src.index("acc = {}")], ns)
    acc = {}
    for k in rows:
        g = ns["grp"](k)
        acc[g] = acc.get(g, 0.0) + k["total_ms"]
    tot = sum(acc.values())
    print(f"### {title} (`profiles/{fname}`: kernel time under rocprofv3, {steps} steps of `{d['command']}`)\n")
    print("| group | ms per step | share |")
    print("|---|---|---|")
    for g, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"| {g} | {v / steps:.2f} | {100 * v / tot:.1f} % |")
    print(f"| total of the listed kernels | {tot / steps:.2f} | |\n")
    print("| kernel | grid class | launches / step | avg ms | GB fetched (FETCH_SIZE × 2) | GB written | MfmaUtil % |")
    print("|---|---|---|---|---|---|---|")
    for k in rows[:22]:
        print(f"| `{k['kernel'][:70]}` | {k.get('class', '')} | {k['launches'] / steps:.1f} | {k['avg_ms']:.3f} | {k.get('fetch_bytes_per_launch', 0) / 1e9:.2f} | "
              f"{k.get('write_bytes_per_launch', 0) / 1e9:.2f} | {k.get('mfma_util_percent', 0):.1f} |")
    print()


def roofline():
    f = P("r06_pmc_traffic.json")
    if not os.path.exists(f):
        return
    print("### Per-kernel roofline of the cfg2 step (`tools/roofline_table.py profiles/r06_pmc_traffic.json`)\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), f, "8"], capture_output=True, text=True)
    print(r.stdout)


def attention():
    f = P("r06_attention_ceiling.jsonl")
    if not os.path.exists(f):
        print("*(r06_attention_ceiling.jsonl missing)*\n")
        return
    print("### Attention forward: the product kernel against its own tile loop without global traffic (`profiles/r06_attention_ceiling.jsonl`)\n")
    print("| variant | B × heads × N | ms | algorithmic TFLOP/s | executed MFMA work as a share of 2.5 PFLOP/s |")
    print("|---|---|---|---|---|")
    for l in open(f):
        if not l.startswith("{"):
            continue
        d = json.loads(l)
        if "error" in d:
            print(f"| {d['variant']} | error | | | |")
            continue
        print(f"| {d['variant']} | {d['B']} × {d['heads']} × {d['N']} | {d['ms']:.4f} | {d['algorithmic_tflops']:.0f} | {100 * d['executed_mfma_share_of_2500']:.1f} % |")
    print()


def valu_rates():
    f = P("r06_valu_rates.jsonl")
    if not os.path.exists(f):
        return
    rows = [json.loads(l) for l in open(f) if l.startswith("{")]
    dev = rows[0]
    by = {}
    for d in rows[1:]:
        by.setdefault(d["case"], {})[d["waves_per_simd"]] = d
    ghz = dev["clock_mhz"] / 1000.0
    print("### What one SIMD needs for the instructions of an attention tile (`profiles/r06_valu_rates.jsonl`, `tools/probe/valu_rates.hip`; "
          f"{dev['cus']} CUs, cycles at the nominal {ghz:.1f} GHz)\n")
    print("One 16x16x32 bf16 MFMA = 16,384 FLOP = 64 scores of a head-dim-64 forward (4 x 64 FLOP per score), i.e. ONE wave-wide instruction of every per-score step.\n")
    print("| instruction stream (16 independent chains per wave) | ns per instruction group per SIMD: 1 / 2 / 4 waves per SIMD | cycles at 4 waves | matrix pipe's share of the 2.5-PFLOP/s peak if this were the whole tile loop (4 waves) |")
    print("|---|---|---|---|")
    for case, w in by.items():
        ns = [w[k]["ns_per_wave_instruction_group_per_simd"] for k in (1, 2, 4)]
        share = w[4]["mfma_share_of_peak_if_this_were_the_tile_loop"]
        print(f"| `{case}` | {ns[0]:.2f} / {ns[1]:.2f} / {ns[2]:.2f} | {ns[2] * ghz:.1f} | {'%.3f' % share if share > 0 else '-'} |")
    print()


def adam_ab():
    f = P("r06_adam_pack_ab.txt")
    if not os.path.exists(f):
        return
    arms = {}
    for l in open(f):
        if l.startswith("UMR_ADAM_PACK="):
            t = l.split()
            arms.setdefault(t[0], []).append(float(t[1]))
    if arms:
        print("### The optimizer launch that writes the packed copies, same-box A/B on the reference recipe (`profiles/r06_adam_pack_ab.txt`)\n")
        print("| arm | images/s (runs, alternating) | median |")
        print("|---|---|---|")
        for a, v in sorted(arms.items()):
            print(f"| `{a}` | {', '.join(f'{x:.1f}' for x in v)} | {statistics.median(v):.1f} |")
        print()


print("<!-- generated by tools/design_tables.py from profiles/r06_*; do not edit by hand -->\n")
lines()
breakdown("r06_pmc_traffic.json", 8, "cfg2 train step by kernel group")
roofline()
attention()
valu_rates()
adam_ab()

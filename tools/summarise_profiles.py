#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>_*) into the committed summaries under profiles/:
   <tag>_kernel_stats*.csv            rocprofv3 --kernel-trace --stats, as written by the profiler
   <tag>_pmc_traffic_ne<ne>_L<L>_n<ranks>.json   HBM bytes per launch and kernel (tools/pmc_table.py)
   <tag>_valu_counters.txt            VALU issue utilisation per kernel
   <tag>_mfma_ab.json                 MFMA counters of the explicit kernel with TMX_MFMA=0 / 1, and its duration both ways
usage: tools/summarise_profiles.py <tag>"""
import collections, csv, json, os, shutil, subprocess, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out"); P = os.path.join(R, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def counters(name):
    path = os.path.join(G, "%s_%s" % (tag, name), "%s_counter_collection.csv" % tag)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


def stats(name):
    path = os.path.join(G, "%s_%s" % (tag, name), "%s_kernel_stats.csv" % tag)
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


for src, dst in (("trace", "kernel_stats"), ("trace_d", "kernel_stats_element_major"), ("trace_mfma1", "kernel_stats_mfma1"), ("trace_ne15", "kernel_stats_ne15"),
                 ("trace_n8", "kernel_stats_rank0_of_8"), ("trace_c4", "kernel_stats_config4_ne30_L40"), ("trace_c5", "kernel_stats_config5_ne120_L60")):
    f = os.path.join(G, "%s_%s" % (tag, src), "%s_kernel_stats.csv" % tag)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, "%s_%s.csv" % (tag, dst)))

for fe, wr, out, what, steps in (("fetch", "write", "ne30_L30_n1", "ne30 np4 L30, 1 GPU", 15),
                                 ("fetch_d", "write_d", "ne30_L30_n1_element_major", "ne30 np4 L30, 1 GPU, TMX_UNIQUE=0 (element-major layout)", 15),
                                 ("fetch_ne15", "write_ne15", "ne15_L30_n1", "ne15 np4 L30 (--ne 15 --ppd 1), 1 GPU", 15),
                                 ("fetch_n8", "write_n8", "ne30_L30_n8", "ne30 np4 L30, share of rank 0 of 8 (tools/rank_share_timing.py 8: "
                                  "lone rank-0 engine of the 8-rank layout, wire left out)", 33),
                                 ("fetch_c4", "write_c4", "ne30_L40_n1", "BASELINE config 4's shape: tools/config4_timing.py, supercell set-up ne30 np4 L40, "
                                  "3 tracers, uniform diffusion, explicit vertical mode, Kessler after every step, 1 GPU", 35),
                                 ("fetch_c5", "write_c5", "ne120_L60_n1", "BASELINE config 5: bench.py --ne 120 --levels 60 --dt 37.5 --physics hs "
                                  "--steps 3 --warmup 1 --develop 2, 1 GPU", 9),
                                 ("fetch_pull", "write_pull", "ne30_L30_n1_hvis_pull", "ne30 np4 L30, 1 GPU, TMX_HVIS_PULL=1 (k_hypervis / k_dss only)", 15)):
    a = os.path.join(G, "%s_%s" % (tag, fe), "%s_counter_collection.csv" % tag)
    b = os.path.join(G, "%s_%s" % (tag, wr), "%s_counter_collection.csv" % tag)
    if os.path.exists(a) and os.path.exists(b):
        dstj = os.path.join(P, "%s_pmc_traffic_%s.json" % (tag, out))
        subprocess.run([sys.executable, os.path.join(R, "tools", "pmc_table.py"), a, b, str(steps), dstj, what], check=True, stdout=subprocess.DEVNULL)
        # stamp the table with the hash of the kernel sources it was taken with (bench.py compares it with the tree it runs from);
        # summarise right after the gpurun call, before the sources are edited again
        sys.path.insert(0, R)
        import bench as _bench
        shaf = os.path.join(G, "%s_%s.sha" % (tag, fe))      # written on the GPU box by tools/profile_round.sh; older passes: TMX_PROFILE_SHA or the current tree
        sha = open(shaf).read().strip() if os.path.exists(shaf) and open(shaf).read().strip() else (os.environ.get("TMX_PROFILE_SHA") or _bench.csrc_sha16())
        tabj = json.load(open(dstj)); tabj["csrc_sha16"] = sha
        json.dump(tabj, open(dstj, "w"), indent=1)

v = counters("valu")
if v:
    with open(os.path.join(P, "%s_valu_counters.txt" % tag), "w") as f:
        f.write("rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- python3 bench.py --steps 5 --warmup 2 "
                "--develop 3 --no-cpu-baseline (ne30 L30, 1 GPU); averages per launch.\n"
                "valu_active/wave_cycles = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both in quad-cycles): the share of a resident wave's "
                "life in which it issues VALU work.\n\n")
        for k, c in sorted(v.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 0])[1]):
            if "SQ_INSTS_VALU" not in c:
                continue
            av = {n: x[1] / x[0] for n, x in c.items()}
            f.write("%-64s launches %4d  INSTS_VALU %12.0f  ACTIVE_INST_VALU %12.0f  WAVE_CYCLES %13.0f  BUSY_CYCLES %12.0f  "
                    "valu_active/wave_cycles %.3f\n" % (k[:64], c["SQ_INSTS_VALU"][0], av["SQ_INSTS_VALU"], av["SQ_ACTIVE_INST_VALU"],
                                                        av["SQ_WAVE_CYCLES"], av["SQ_BUSY_CYCLES"],
                                                        av["SQ_ACTIVE_INST_VALU"] / max(av["SQ_WAVE_CYCLES"], 1.0)))

ab = {"method": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE and a separate "
                "--kernel-trace --stats run, each with TMX_MFMA=0 and TMX_MFMA=1, of `python3 bench.py --steps 5 --warmup 2 --develop 3 "
                "--no-cpu-baseline` (ne30 np4 L30, 1 GPU, element-major layout on both sides); per launch of the fused explicit kernel k_h_explicit<true,NT,MF>, summed "
                "over its four instantiations of an ARS343 step (NT = 0, 4, 6, 7 combination terms).  mfma busy fraction = "
                "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the share of SIMD-cycles the matrix unit is busy."}
for env, cn, tn in (("TMX_MFMA=0", "mfma0", "trace_d"), ("TMX_MFMA=1", "mfma1", "trace_mfma1")):
    # (both sides on the element-major layout, TMX_UNIQUE=0: the matrix-unit variant has no node-unique form)
    c = counters(cn); st = stats(tn) or stats("trace")
    if not c:
        continue
    ex = {k: x for k, x in c.items() if "k_h_explicit" in k}
    tot = lambda n: sum(x[n][1] / x[n][0] for x in ex.values() if n in x)
    busy, gui, mops = tot("SQ_VALU_MFMA_BUSY_CYCLES"), tot("GRBM_GUI_ACTIVE"), tot("SQ_INSTS_VALU_MFMA_MOPS_F64")
    dur = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in st if "k_h_explicit" in r["Name"] or "k_h_w_update" in r["Name"]}
    step_ns = sum(float(r["TotalDurationNs"]) for r in st if not r["Name"].startswith("__amd")) if st else 0.0
    ab[env] = {"explicit_kernel_us_per_launch": dur, "explicit_kernels_us_per_step": sum(v for k, v in dur.items() if "k_h_explicit" in k),
               "mfma_instructions_per_step": mops, "mfma_busy_cycles_per_step": busy,
               "mfma_busy_fraction_of_explicit_kernel": busy / max(gui / 8.0 * 1024.0, 1.0),
               "all_kernels_us_per_profiled_run": step_ns / 1e3}
if len(ab) > 1:
    json.dump(ab, open(os.path.join(P, "%s_mfma_ab.json" % tag), "w"), indent=1)
print(sorted(f for f in os.listdir(P) if f.startswith(tag)))

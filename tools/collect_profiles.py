#!/usr/bin/env python3
"""Turn gpurun_out/prof (tools/refresh_profiles.sh) into the committed summaries under profiles/.

usage: python tools/collect_profiles.py rNN"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (source_hash)

tag = sys.argv[1]
pmc_only = "--pmc-only" in sys.argv[2:]      # on the GPU box, between the counter passes and the bench runs of one refresh
src, dst = "gpurun_out/prof", "profiles"


def find(name):
    f = glob.glob(os.path.join(src, "**", name), recursive=True)
    return f[0] if f else None


def pmc(name):
    f = find(name + "_counter_collection.csv")
    if not f:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(calls[k]) for c, v in d.items()} for k, d in agg.items() if "sgpr" in k}


def trace_avg_us(name, is_kernel):
    """average duration (us) of the kernels `is_kernel(name)` accepts in the kernel trace of one --pmc pass"""
    f = find(name + "_kernel_trace.csv")
    if not f:
        return None
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if is_kernel(r["Kernel_Name"])]
    return sum(d) / len(d) if d else None


def is_embed(kname):
    return ("embed_kernel" in kname or "embed_big_kernel" in kname) and "redo" not in kname


def kernel_stats(name, out, header):
    f = find(name + "_kernel_stats.csv")
    if not f:
        return
    with open(os.path.join(dst, out), "w") as o:
        o.write("# " + header + "  (durations in us)\n")
        for r in csv.DictReader(open(f)):
            o.write("%-100s calls %5s  avg %9.1f  min %9.1f  max %9.1f  %5s%%\n" % (
                r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                r["Percentage"]))
    print(open(os.path.join(dst, out)).read())


def bench_line(name, out):
    f = os.path.join(src, name)
    if not os.path.exists(f):
        return
    lines = [l for l in open(f).read().strip().splitlines() if l.startswith("{")]
    if lines:
        json.loads(lines[-1])
        open(os.path.join(dst, out), "w").write(lines[-1] + "\n")


if not pmc_only:
    kernel_stats("kt", tag + "_kernel_stats.txt", "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-end-to-end")
    kernel_stats("kt_stress", tag + "_stress_kernel_stats.txt", "rocprofv3 --kernel-trace --stats -- python bench.py --workload stress --no-cpu-baseline --steps 50")
    kernel_stats("kt_pairs128", tag + "_pairs128_kernel_stats.txt", "rocprofv3 --kernel-trace --stats -- python bench.py --workload pairs128 --no-cpu-baseline --steps 50")
    kernel_stats("kt_consumers", tag + "_consumers_kernel_stats.txt", "rocprofv3 --kernel-trace --stats -- python tools/run_f1.py 3")
    kernel_stats("kt_pairlist", tag + "_pairlist_kernel_stats.txt", "rocprofv3 --kernel-trace --stats -- python bench.py --workload pairlist --no-cpu-baseline --steps 50")
    kernel_stats("kt_wide", tag + "_wide_kernel_stats.txt", "rocprofv3 --kernel-trace --stats -- python tools/run_embed.py kitti00 5 8192   "
                 "(debug bit 13: every graph on the wide-range instance - three bf16 planes, 24-bit operands)")
    for nm, out in (("seq_parity_world.txt", "_seq_parity_world.txt"), ("seq_parity_both.txt", "_seq_parity.txt"),
                    ("tailmix_probe.txt", "_tailmix_probe.txt"), ("any_shape.txt", "_any_shape.txt")):
        f = os.path.join(src, nm)
        if os.path.exists(f):
            open(os.path.join(dst, tag + out), "w").write(open(f).read())
    try:
        line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
        wide = json.loads(line)["roofline"].get("wide_range")
        if wide:
            json.dump({"source": "python bench.py (the default line's roofline.wide_range)", "wide_range": wide,
                       "default_ms_per_step": json.loads(line)["ms_per_step"], "default_value": json.loads(line)["value"]},
                      open(os.path.join(dst, tag + "_wide_bench.json"), "w"), indent=1)
    except (OSError, IndexError, KeyError, ValueError):
        pass
    for nm, out, head in (
            ("fuzz_f1.txt", "_fuzz_f1.txt", "# python tools/exp/fuzz_f1_one_call.py <seed> 120, seeds 41 42 43: sgpr_f1_max against the sorted host computation\n"),
            ("fuzz_anyshape_wide.txt", "_any_shape_fuzz_wide.txt", "# python tools/exp/fuzz_anyshape.py 40 wide: random architectures inside sgpr_wide.hip's "
             "limits (matrix-core any-shape embed + dense tail) against the oracle\n"),
            ("f1_scan_timeline.txt", "_f1_scan_timeline.txt", "# SGPR_HIP_LIB=variants/libsgpr_stamps.so (tools/build_variant.sh stamps -DSGPR_F1_SCAN_STAMPS=1) "
             "python tools/exp/f1_scan_timeline.py kitti|world\n# per-wave time stamps of sgpr_f1_max's first pass (4096 waves), us from the first wave's start; "
             "before this round's changes (HISTORY 14):\n# streamed median 17 / max 25, classified p90 35 / max 53, flushed p90 49 / max 52, kernel 54 us\n"),
            ("embed_timeline.txt", "_embed_timeline.txt", "# SGPR_HIP_LIB=variants/libsgpr_estamps.so (tools/build_variant.sh estamps -DSGPR_EMBED_STAMPS=1) "
             "python tools/exp/embed_timeline.py  (KITTI-00 shape, ordered launch)\n")):
        f = os.path.join(src, nm)
        if os.path.exists(f) and os.path.getsize(f) > 0:
            open(os.path.join(dst, tag + out), "w").write(head + open(f).read())
    f = os.path.join(src, "plain_embed.txt")
    if os.path.exists(f):
        with open(os.path.join(dst, tag + "_plain_embed.txt"), "w") as o:
            o.write("# python tools/run_auto.py 50: sgpr_embed without a node_cap promise (64-row launch + hand-over of larger graphs) against\n"
                    "# the capped and the ordered launch of the same graphs; host-timed loops of 50 calls (us per call), bitwise = same pooled vectors\n")
            o.write("".join(l for l in open(f) if "amdgpu.ids" not in l))
    try:
        line = [l for l in open(os.path.join(src, "bench_cpu100k.json")) if l.startswith("{")][-1]
        cb = json.loads(line)["cpu_baseline"]
        json.dump({"source": "python bench.py --cpu-pairs 100000 --steps 20 --no-end-to-end --no-wide-range (SURVEY.md 8d's CPU sample)",
                   "cpu_baseline": cb}, open(os.path.join(dst, tag + "_cpu_baseline_100k.json"), "w"), indent=1)
    except (OSError, IndexError, KeyError, ValueError):
        pass
    for kind in ("kitti", "world"):
        f = os.path.join(src, "f1_phases_%s.log" % kind)
        if os.path.exists(f):
            with open(os.path.join(dst, tag + "_f1_phases_%s.txt" % kind), "w") as o:
                o.write("# python tools/f1_phases.py %s  (sgpr_f1_max on a KITTI-00-sized matrix)\n" % kind)
                o.write("".join(l for l in open(f) if "amdgpu.ids" not in l))
    if os.path.exists(os.path.join(src, "consumers.log")):
        with open(os.path.join(dst, tag + "_consumers.txt"), "w") as o:
            o.write("# python tools/run_f1.py 10 check  (KITTI-00-sized matrix of the bench, device F1-max / ROC area)\n")
            o.write("".join(l for l in open(os.path.join(src, "consumers.log")) if "amdgpu.ids" not in l))
    cons = {}
    for nm in ("sq_consumers", "FETCH_SIZE_consumers", "WRITE_SIZE_consumers"):
        for kname, d in pmc(nm).items():
            if "pair_" in kname or "slab" in kname or "f1_" in kname:
                cons.setdefault(kname, {}).update(d)
    if cons:
        with open(os.path.join(dst, tag + "_consumers_pmc.txt"), "w") as o:
            o.write("# rocprofv3 --pmc (separate passes) -- python tools/run_f1.py 1; per-launch averages over the passes with and without the\n"
                    "# ranking; SQ counters in millions, FETCH_SIZE / WRITE_SIZE in KiB (the matrix read is 80 549 KiB)\n")
            for kname, d in cons.items():
                o.write("%-40s %s\n" % (kname[:40], {c: (round(v, 1) if "SIZE" in c else round(v / 1e6, 3)) for c, v in sorted(d.items())}))
        print(open(os.path.join(dst, tag + "_consumers_pmc.txt")).read())
    for name, out in (("bench.json", "_bench.json"), ("bench_under_rocprof.json", "_bench_under_rocprof.json"),
                      ("bench_stress.json", "_stress_bench.json"), ("bench_pairs128.json", "_pairs128_bench.json"),
                      ("bench_kitti5seq.json", "_kitti5seq_bench.json"), ("bench_gloo2.json", "_gloo2ranks_one_gpu_bench.json"),
                      ("bench_pairlist.json", "_pairlist_bench.json"),
                      ("bench_gloo2_pairlist.json", "_gloo2ranks_one_gpu_pairlist_bench.json")):
        bench_line(name, tag + out)

hbm = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python "
                 "tools/run_embed.py <shape> 3; per-launch averages (" + tag + ")",
       "source_hash": bench.source_hash(),
       "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB. WRITE_SIZE matches the known output bytes (pooled: 4541 x 128 B = "
                    "567.6 KiB). FETCH_SIZE under-reports on gfx950 (MI355X_MICROARCH.md: exactly 1/2 for 16 B/lane "
                    "streams); the corrected read traffic is bracketed by [raw, 2 x raw] - bench.py uses 2 x raw."}
shapes = {"kitti00": (4541, 100, 10), "stress": (2048, 256, 20), "pairs128": (256, 64, 10)}
for shape, (g, n, k) in shapes.items():
    fetch, write = pmc("fetch_" + shape), pmc("write_" + shape)
    for kname in fetch:
        e = {"FETCH_SIZE_KiB": round(fetch[kname].get("FETCH_SIZE", 0.0), 1),
             "WRITE_SIZE_KiB": round(write.get(kname, {}).get("WRITE_SIZE", 0.0), 1)}
        key = kname.split("<")[0]
        if "embed" in kname:
            e.update({"graphs_per_launch": g, "node_num": n, "K": k})
        if shape == "kitti00":
            hbm[key] = e
        else:
            hbm.setdefault(shape, {})[key] = e
# instruction mix / issue counters of the dominant kernel (bench.py's roofline.issue), same source hash
for shape in shapes:
    ins = {}
    for name in ("sq1_", "sq2_", "sq3_"):
        for kname, d in pmc(name + shape).items():
            if is_embed(kname):
                ins.update({c: round(v) for c, v in d.items()})
                ins["kernel"] = kname
    # the effective shader clock of THIS kernel under load, two ways (MI355X_MICROARCH.md, DVFS): SQ_BUSY_CYCLES is summed over
    # the 32 shader engines and counts only while waves are resident (a lower bound: an engine that runs dry early stops
    # counting), GRBM_GUI_ACTIVE over the 8 XCDs with the collection's own start / stop inside (an upper bound); each
    # divided by the kernel's duration in the pass that read it
    t1, t3 = trace_avg_us("sq1_" + shape, is_embed), trace_avg_us("sq3_" + shape, is_embed)
    if ins.get("SQ_BUSY_CYCLES") and t1:
        ins["duration_us_sq1_pass"] = round(t1, 2)
        ins["clock_ghz_from_sq_busy"] = round(ins["SQ_BUSY_CYCLES"] / 32.0 / (t1 * 1e3), 4)
    if ins.get("GRBM_GUI_ACTIVE") and t3:
        ins["duration_us_sq3_pass"] = round(t3, 2)
        ins["clock_ghz_from_grbm"] = round(ins["GRBM_GUI_ACTIVE"] / 8.0 / (t3 * 1e3), 4)
    if ins:
        (hbm if shape == "kitti00" else hbm.setdefault(shape, {}))["embed_kernel_counters"] = ins
    tail = {}
    for name in ("sq1_", "sq2_"):
        for kname, d in pmc(name + shape).items():
            if "score_all_pairs_kernel" in kname:
                tail.update({c: round(v) for c, v in d.items()})
                tail["kernel"] = kname
    if tail and shape == "kitti00":
        hbm["tail_kernel_counters"] = tail
json.dump(hbm, open(os.path.join(dst, "pmc_hbm_latest.json"), "w"), indent=2)
if pmc_only:
    print("pmc_hbm_latest.json written for source hash", hbm["source_hash"])
    sys.exit(0)
with open(os.path.join(dst, tag + "_pmc_sq.txt"), "w") as f:
    f.write("# rocprofv3 --pmc (separate passes) on tools/run_embed.py <shape> 3; per-launch averages in millions (" + tag + ")\n")
    for shape in shapes:
        for name in ("sq1_", "sq2_", "sq3_"):
            for kname, d in pmc(name + shape).items():
                f.write("%-9s %-40s %s\n" % (shape, kname[:40], {c: round(v / 1e6, 3) for c, v in sorted(d.items())}))
print(json.dumps(hbm, indent=1)[:1500])
print(open(os.path.join(dst, tag + "_pmc_sq.txt")).read())

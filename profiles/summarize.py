#!/usr/bin/env python3
"""Condense rocprofv3 output directories into small summaries (run on the GPU box by profiles/collect.sh; the raw
per-dispatch CSVs are tens of MB and are deleted afterwards).
  kernel_stats.csv      : per kernel (template arguments kept): calls, total ms, average us, share
  pmc_traffic.csv       : per kernel: launches, FETCH_SIZE / WRITE_SIZE per launch as reported (KB -> MB), and the
                          gfx950-corrected HBM bytes per launch (FETCH x 2 for wide coalesced reads + WRITE;
                          MI355X_MICROARCH.md, section HBM)
  timed_region_family_union.csv (round 6): per kernel family and stage: sum of the launches' durations, the UNION of their intervals
                          (launches of a family overlap on two streams), algorithmic GB, GB/s and fraction of 8 TB/s by both
  excess_by_kernel.csv (round 6): per (stage, variant) of the timed region: floor_us = max(bytes / 8 TB/s, flops / 2.5 PFLOP/s) of one
                          launch, the side that binds, excess_ms = calls x (avg_us - floor_us), sorted by excess
  timed_region_kernel_stats.csv (round 4): the dispatches BETWEEN bench.py's two marker kernels (imk_mark_kernel with 64 and
                          128 work-items: the timed region), per kernel variant: calls, average us from rocprofv3's own begin /
                          end timestamps, and -- joined from the bench line of the traced run (bench_traced.json:
                          timed_region_kernel_totals, the library's own sums over every launch) -- algorithmic MB and GFLOP per
                          launch, hence GB/s, TFLOP/s and the fraction of the 8 TB/s / 2.5 PFLOP/s peak without any HIP-event bracket
"""
import collections
import csv
import glob
import json
import os
import re
import sys

HBM_PEAK_GBS, MFMA_PEAK_TFLOPS = 8000.0, 2500.0


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-zA-Z_0-9]+?)(I|E)", n)
    if m:
        return m.group(1)
    return n.split("(")[0][:90]


def main(out):
    stats = glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(stats[0])):
            k = short(r["Name"])
            agg[k][0] += int(r["Calls"])
            agg[k][1] += float(r["TotalDurationNs"])
        tot = sum(v[1] for v in agg.values())
        with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, v[0], round(v[1] / 1e6, 3), round(v[1] / v[0] / 1e3, 3), round(100 * v[1] / tot, 2)])
    timed_region(out)
    tr = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(out, f"pmc_{cname}", "*", "*_counter_collection.csv"))
        if not files:
            continue
        rows = list(csv.DictReader(open(files[0])))
        # only the generations (warm-up, timed, exclusive pass): everything from the first fused-IM launch on -- the
        # 1800 pre-training steps of the synthetic ensemble before it would swamp the launch mix of the timed region
        # (the first ensemble forward starts with the inference-only stem-on-load kernel conv_pipe_kernel<6, ...>; before
        # round 2 the marker was the first fused-IM launch and the uint8 stem conv_pipe_kernel<4, ...> in front of it)
        first_gen = min((int(r["Dispatch_Id"]) for r in rows if "conv_pipe_kernel<6" in r["Kernel_Name"]), default=None)
        if first_gen is None:
            first = min((int(r["Dispatch_Id"]) for r in rows if "im_binary" in r["Kernel_Name"] or "head_im" in r["Kernel_Name"]), default=0)
            first_gen = max((int(r["Dispatch_Id"]) for r in rows if int(r["Dispatch_Id"]) < first and "conv_pipe_kernel<4" in r["Kernel_Name"]),
                            default=0)
        for r in rows:
            if int(r["Dispatch_Id"]) < first_gen:
                continue
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            d = tr.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n_FETCH_SIZE": 0, "n_WRITE_SIZE": 0})
            d[cname] += float(r["Counter_Value"])
            d["n_" + cname] += 1
    if tr:
        with open(os.path.join(out, "pmc_traffic.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "workgroups", "launches", "fetch_MB_per_launch_raw", "write_MB_per_launch",
                        "hbm_MB_per_launch_corrected(2*fetch+write)"])
            for k, d in sorted(tr.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
                nf, nw = max(d["n_FETCH_SIZE"], 1), max(d["n_WRITE_SIZE"], 1)
                fe, wr = d["FETCH_SIZE"] / nf / 1024, d["WRITE_SIZE"] / nw / 1024      # counters are in KB
                w.writerow([k[0], k[1], nf, round(fe, 3), round(wr, 3), round(2 * fe + wr, 3)])
    for big in glob.glob(os.path.join(out, "*", "*", "*_kernel_trace.csv")) + \
            glob.glob(os.path.join(out, "*", "*", "*_counter_collection.csv")):
        os.remove(big)


def timed_region(out):
    traces = glob.glob(os.path.join(out, "trace", "*", "*_kernel_trace.csv"))
    if not traces:
        return
    rows = []
    for f in traces:     # one file per process; the bench is the one with the markers
        rr = list(csv.DictReader(open(f)))
        if any("imk_mark_kernel" in r["Kernel_Name"] for r in rr):
            rows = rr
            break
    marks = {}
    stage_marks = []     # (start timestamp, stage that begins there): bench.py's marker 4 opens a generation's inference stage, 3 its epoch
    for r in rows:
        if "imk_mark_kernel" in r["Kernel_Name"]:
            ident = int(r["Grid_Size_X"]) // 64
            if ident in (1, 2):
                marks[ident] = r
            elif ident in (3, 4):
                stage_marks.append((int(r["Start_Timestamp"]), "training" if ident == 3 else "inference"))
    if 1 not in marks or 2 not in marks:
        return
    t0, t1 = int(marks[1]["End_Timestamp"]), int(marks[2]["Start_Timestamp"])
    stage_marks = sorted(m for m in stage_marks if t0 <= m[0] <= t1)
    import bisect
    starts = [m[0] for m in stage_marks]

    def stage_of(ts):
        i = bisect.bisect_right(starts, ts) - 1
        return stage_marks[i][1] if i >= 0 else ("inference" if stage_marks else "")
    agg = collections.defaultdict(lambda: [0, 0.0])
    spans = collections.defaultdict(list)      # (stage, family) -> [(start, end)]: the family's launches overlap (two ensemble members on two streams)
    for r in rows:
        if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1 and "imk_mark_kernel" not in r["Kernel_Name"]:
            k = (stage_of(int(r["Start_Timestamp"])), short(r["Kernel_Name"]))
            agg[k][0] += 1
            agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            fam = k[1].split("<")[0]
            fam = "conv_pipe_kernel+conv_wide_kernel" if fam in ("conv_pipe_kernel", "conv_wide_kernel") else fam      # one family in the bench's hook
            spans[(k[0], fam)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k[1]))
    totals, stage_totals = {}, {}
    try:
        # round 6: the full record of the traced run sits beside its (now compact) stdout line (bench.py --detail)
        det = os.path.join(out, "bench_traced_detail.json")
        rec = json.load(open(det)) if os.path.exists(det) else json.loads(
            [l for l in open(os.path.join(out, "bench_traced.json")).read().splitlines() if l.startswith("{")][-1])
        totals = rec.get("timed_region_kernel_totals", {})
        stage_totals = rec.get("stage_kernel_totals_per_generation", {})
    except Exception:
        pass
    # the library sums some families under one name (wgf_stage1 + wgf_stage2, the step tail, the conv_mfma variants by tile): those
    # rows carry no bytes here; conv_pipe / conv_wide (the dominant family) match by their full template names.  A variant that runs
    # in both stages (the deep levels' kernels) has other bytes per launch in each (256 images per inference call, 32 per step): the
    # bench line's per-stage totals (one generation outside the clock) price it per stage.
    tot_ns = sum(v[1] for v in agg.values())
    stage_ns = collections.defaultdict(float)
    for (st, _), v in agg.items():
        stage_ns[st] += v[1]
    with open(os.path.join(out, "timed_region_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["stage", "kernel", "calls", "total_ms", "avg_us", "percent_of_kernel_time", "percent_of_stage_kernel_time",
                    "algorithmic_MB_per_launch", "GFLOP_per_launch", "GBps", "frac_of_8TBps", "TFLOPs", "frac_of_2.5PFLOPs", "region_ms"])
        for (st, k), v in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
            avg_us = v[1] / v[0] / 1e3
            t = stage_totals.get(st, {}).get(k) or totals.get(k)      # exact kernel name only: a family-wide average would misprice its variants
            if t and t.get("launches"):
                mb, gf = t["MB_per_launch"], t["GFLOP_per_launch"]
                gbps, tf = mb / avg_us * 1e3, gf / avg_us * 1e3
                extra = [mb, gf, round(gbps, 1), round(gbps / HBM_PEAK_GBS, 4), round(tf, 1), round(tf / MFMA_PEAK_TFLOPS, 4)]
            else:
                extra = ["", "", "", "", "", ""]
            w.writerow([st, k, v[0], round(v[1] / 1e6, 3), round(avg_us, 3), round(100 * v[1] / max(tot_ns, 1), 2),
                        round(100 * v[1] / max(stage_ns[st], 1), 2)] + extra + [round((t1 - t0) / 1e6, 3)])
    # Per family and stage: the SUM of the launches' durations against the time during which at least one of them was running (the
    # union of their intervals).  Launches of one family overlap -- the two ensemble members' forwards run on two streams, a training
    # step's weight gradients beside its dgrads -- so bytes / sum-of-durations prices every launch as if it had the chip to itself and
    # took that long: the bandwidth the family DELIVERED while it ran is bytes / union.
    def union_ns(iv):
        iv = sorted((a, b) for a, b, _ in iv)
        tot, cs, ce = 0, None, None
        for a, b in iv:
            if ce is None or a > ce:
                if ce is not None:
                    tot += ce - cs
                cs, ce = a, b
            else:
                ce = max(ce, b)
        return tot + (ce - cs if ce is not None else 0)
    with open(os.path.join(out, "timed_region_family_union.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["stage", "family", "calls", "sum_ms", "union_ms", "overlap_factor", "algorithmic_GB", "GFLOP", "GBps_by_sum", "GBps_by_union",
                    "frac_of_8TBps_by_sum", "frac_of_8TBps_by_union", "TFLOPs_by_union", "frac_of_2.5PFLOPs_by_union"])
        def fam_rows(stage_filter, label):
            out_rows = []
            fams = sorted({fk[1] for fk in spans})
            for fam in fams:
                iv = [x for (st, fm), lst in spans.items() if fm == fam and (stage_filter is None or st == stage_filter) for x in lst]
                if not iv:
                    continue
                mb = gf = 0.0
                priced = True
                for a, b, name in iv:
                    st = stage_of(a)
                    t = stage_totals.get(st, {}).get(name) or totals.get(name)
                    if t and t.get("launches"):
                        mb += t["MB_per_launch"]; gf += t["GFLOP_per_launch"]
                    else:
                        priced = False
                sm, un = sum(b - a for a, b, _ in iv) / 1e6, union_ns(iv) / 1e6
                if priced and un > 0:
                    out_rows.append([label, fam, len(iv), round(sm, 3), round(un, 3), round(sm / un, 3), round(mb / 1e3, 3), round(gf, 1),
                                     round(mb / sm, 1), round(mb / un, 1), round(mb / sm / HBM_PEAK_GBS, 4), round(mb / un / HBM_PEAK_GBS, 4),
                                     round(gf / un, 1), round(gf / un / MFMA_PEAK_TFLOPS, 4)])
                else:
                    out_rows.append([label, fam, len(iv), round(sm, 3), round(un, 3), round(sm / max(un, 1e-9), 3)] + [""] * 8)
            return sorted(out_rows, key=lambda r: -r[3])
        for lab, flt in (("all", None), ("inference", "inference"), ("training", "training")):
            w.writerows(fam_rows(flt, lab))
    # Per variant: the floor of ONE launch = max(algorithmic bytes / 8 TB/s, flops / 2.5 PFLOP/s), which side binds, and the time the
    # variant spends ABOVE that floor in the timed region (excess_ms = calls x (avg_us - floor_us)); sorted by excess: the rows at the
    # top own the gap between the step and its roofline.  Kernels the library does not price (tiny per-channel reductions, the step
    # tail: KB-sized) have floor 0: all of their time is excess (launch latency).
    rows = []
    for (st, k), v in agg.items():
        avg_us = v[1] / v[0] / 1e3
        t = stage_totals.get(st, {}).get(k) or totals.get(k)
        if t and t.get("launches"):
            fb, ff = t["MB_per_launch"] * 1e6 / (HBM_PEAK_GBS * 1e9) * 1e6, t["GFLOP_per_launch"] * 1e9 / (MFMA_PEAK_TFLOPS * 1e12) * 1e6
            floor, bound, mb, gf = max(fb, ff), ("mfma" if ff > fb else "hbm"), t["MB_per_launch"], t["GFLOP_per_launch"]
        else:
            floor, bound, mb, gf = 0.0, "unpriced", "", ""
        rows.append([st, k, v[0], round(v[1] / 1e6, 3), round(avg_us, 3), mb, gf, round(floor, 3), bound,
                     round(floor / avg_us, 4) if avg_us else "", round(v[0] * (avg_us - floor) / 1e3, 3)])
    rows.sort(key=lambda r: -r[-1])
    tot_excess = sum(r[-1] for r in rows)
    with open(os.path.join(out, "excess_by_kernel.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["stage", "kernel", "calls", "total_ms", "avg_us", "algorithmic_MB_per_launch", "GFLOP_per_launch", "floor_us",
                    "binding_side", "floor_over_avg", "excess_ms", "percent_of_excess"])
        for r in rows:
            w.writerow(r + [round(100 * r[-1] / max(tot_excess, 1e-9), 2)])
        w.writerow(["TOTAL", "", sum(r[2] for r in rows), round(sum(r[3] for r in rows), 3), "", "", "", "", "", "", round(tot_excess, 3), 100.0])


if __name__ == "__main__":
    main(sys.argv[1])

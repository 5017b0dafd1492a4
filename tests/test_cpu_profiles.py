"""The evidence pipeline itself (profiles/summarize.py), on a synthetic rocprofv3 kernel trace: the timed region is cut at bench.py's marker
dispatches, stages at the stage markers, every (stage, variant) gets its floor max(bytes / 8 TB/s, flops / 2.5 PFLOP/s) and its excess,
and a family's overlapping launches are priced both by the sum of their durations and by the union of their intervals."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_summarize_cuts_prices_and_unions(tmp_path):
    d = tmp_path / "trace" / "x"
    d.mkdir(parents=True)
    rows = []

    def k(name, gx, a, b):
        rows.append({"Kernel_Name": name, "Grid_Size_X": gx, "Start_Timestamp": a, "End_Timestamp": b})
    us = 1000
    k("void (anonymous namespace)::conv_pipe_kernel<3, 1>(A)", 1024, 1 * us, 5 * us)            # before the region: ignored
    k("imk_mark_kernel(int)", 64, 9 * us, 10 * us)                                                # marker 1: region opens
    k("imk_mark_kernel(int)", 256, 11 * us, 12 * us)                                              # marker 4: inference stage
    k("void (anonymous namespace)::conv_pipe_kernel<3, 1>(A)", 1024, 100 * us, 1100 * us)         # two overlapping launches (two streams)
    k("void (anonymous namespace)::conv_pipe_kernel<3, 1>(A)", 1024, 600 * us, 1600 * us)
    k("void (anonymous namespace)::conv_wide_kernel<1>(A)", 1024, 2000 * us, 2500 * us)
    k("imk_mark_kernel(int)", 192, 2600 * us, 2601 * us)                                          # marker 3: training stage
    k("void (anonymous namespace)::conv_gemm_kernel<0, true>(A, B)", 1024, 3000 * us, 3400 * us)
    k("_ZN12_GLOBAL__N_118bn_finalize_kernelEPKfiPf", 64, 3400 * us, 3405 * us)                   # a name the demangler gave up on
    k("imk_mark_kernel(int)", 128, 5000 * us, 5001 * us)                                          # marker 2: region closes
    k("void (anonymous namespace)::conv_pipe_kernel<3, 1>(A)", 1024, 6000 * us, 7000 * us)        # after the region: ignored
    with open(d / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    totals = {"conv_pipe_kernel<3, 1>": {"launches": 2, "MB_per_launch": 4000.0, "GFLOP_per_launch": 100.0},     # byte floor 500 us
              "conv_wide_kernel<1>": {"launches": 1, "MB_per_launch": 800.0, "GFLOP_per_launch": 1000.0},         # flop floor 400 us
              "conv_gemm_kernel<0, true>": {"launches": 1, "MB_per_launch": 1600.0, "GFLOP_per_launch": 10.0},    # byte floor 200 us
              "bn_finalize_kernel": {"launches": 1, "MB_per_launch": 0.008, "GFLOP_per_launch": 0.0}}
    (tmp_path / "bench_traced_detail.json").write_text(json.dumps({"timed_region_kernel_totals": totals, "stage_kernel_totals_per_generation": {}}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "summarize.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tr = {(x["stage"], x["kernel"]): x for x in csv.DictReader(open(tmp_path / "timed_region_kernel_stats.csv"))}
    assert set(tr) == {("inference", "conv_pipe_kernel<3, 1>"), ("inference", "conv_wide_kernel<1>"), ("training", "conv_gemm_kernel<0, true>"),
                       ("training", "bn_finalize_kernel")}
    p = tr[("inference", "conv_pipe_kernel<3, 1>")]
    assert int(p["calls"]) == 2 and float(p["avg_us"]) == 1000.0 and float(p["GBps"]) == 4000.0 and float(p["frac_of_8TBps"]) == 0.5
    ex = {(x["stage"], x["kernel"]): x for x in csv.DictReader(open(tmp_path / "excess_by_kernel.csv"))}
    e = ex[("inference", "conv_pipe_kernel<3, 1>")]
    assert float(e["floor_us"]) == 500.0 and e["binding_side"] == "hbm" and abs(float(e["excess_ms"]) - 1.0) < 1e-9
    e = ex[("inference", "conv_wide_kernel<1>")]
    assert float(e["floor_us"]) == 400.0 and e["binding_side"] == "mfma" and abs(float(e["excess_ms"]) - 0.1) < 1e-9
    e = ex[("training", "conv_gemm_kernel<0, true>")]
    assert float(e["floor_us"]) == 200.0 and e["binding_side"] == "hbm" and float(e["floor_over_avg"]) == 0.5
    assert list(ex)[0] == ("inference", "conv_pipe_kernel<3, 1>")                               # sorted by excess
    assert abs(float(ex[("TOTAL", "")]["excess_ms"]) - (1.0 + 0.1 + 0.2 + 0.005)) < 1e-3
    un = {(x["stage"], x["family"]): x for x in csv.DictReader(open(tmp_path / "timed_region_family_union.csv"))}
    u = un[("inference", "conv_pipe_kernel+conv_wide_kernel")]
    # 1000 + 1000 + 500 us of launches, of which 500 us overlap: union 2000 us; 4000 + 4000 + 800 MB
    assert float(u["sum_ms"]) == 2.5 and float(u["union_ms"]) == 2.0 and float(u["overlap_factor"]) == 1.25
    assert float(u["GBps_by_sum"]) == 3520.0 and float(u["GBps_by_union"]) == 4400.0 and float(u["frac_of_8TBps_by_union"]) == 0.55
    assert float(un[("training", "conv_gemm_kernel")]["overlap_factor"]) == 1.0
    assert ("all", "conv_pipe_kernel+conv_wide_kernel") in un
    assert not any(n.endswith("_kernel_trace.csv") for _, _, fs in os.walk(tmp_path) for n in fs)   # the per-dispatch trace is not kept

#!/usr/bin/env python3
"""Refresh the generated tables of DESIGN.md (section 3 kernel table, section 5 share table, section 6 measurement table)
from the bench lines under profiles/.  usage: python tools/design_update.py"""
import json
import re
import subprocess


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    s = open("DESIGN.md").read()
    b = line("profiles/r06_bench_final.json")
    tab = subprocess.run(["python", "tools/design_table.py", "profiles/r06_bench_final.json"], capture_output=True, text=True).stdout.strip()
    s = re.sub(r"\| kernel \| launches / step \|.*?whole-step fraction [0-9.]+", lambda m: tab, s, flags=re.S)
    T1 = b["ms_per_step"]
    rows = []
    for n in (2, 4, 8):
        sh = line("profiles/r06_bench_one_ranks_share_of_%d_gpus.json" % n)
        rows.append("| %d | %.2f ms (median step %.2f) | %.2f× | %.3f |" % (n, sh["ms_per_step"], sh["step_ms"]["median"], T1 / sh["ms_per_step"], sh["roofline_step"]["frac"]))
    share = ("| N | one rank's share (`profiles/r06_bench_one_ranks_share_of_N_gpus.json`) | ceiling T₁ / share (T₁ = %.2f ms) | whole-step roofline fraction of the share |\n|---|---|---|---|\n" % T1) + "\n".join(rows)
    s = re.sub(r"\| N \| one rank's share \(`profiles/r06_bench_one_ranks_share_of_N_gpus\.json`\).*?\n\| 8 \|[^\n]*", lambda m: share, s, flags=re.S)
    oc, cb = b["other_configs"], b["cpu_baseline"]
    na, npo = line("profiles/r06_bench_final_no_acting.json"), line("profiles/r06_bench_final_no_policy_outputs.json")
    old, old_na = line("profiles/r06_bench_before_mid_tile_final.json"), line("profiles/r06_bench_before_mid_tile_final_no_acting.json")

    def row(name, o):
        r, c = o["roofline"], o.get("cpu_baseline", {})
        return "| %s | **%.3f** | %.1f | %.1f k | %.2f of 8 TB/s (%.1f MB in %.1f µs) | %.0f transitions/s (1 thread, full batch) → %.0f× |" % (
            name, o["ms_per_step"], o["learner_steps_per_sec"], o["value"] / 1e3, r["frac"], r["algorithmic_bytes_per_launch"] / 1e6,
            r["avg_launch_ms"] * 1e3, c.get("value", 0), o.get("speedup_vs_cpu_baseline", 0))
    r = b["roofline"]
    mt = "| config (1× MI355X, f32 results) | ms / step | learner steps/s | sampled transitions/s | frame gather | CPU baseline on the box's host |\n|---|---|---|---|---|---|\n"
    mt += "| **IQN-LSTM B=512 T=80, burn-in 40, n=2 (headline; the actor's q-values stored like the reference)** | **%.2f** (median %.2f, p10 %.2f, p90 %.2f) | **%.2f** | **%.1f k** | **%.3f of 8 TB/s** (3.526 GB in %.3f ms; %.2f of the best plain copy on the box) | %.0f transitions/s (1 thread; linearity check %.2f) → **%.0f×**; 32 threads: %.0f |\n" % (
        b["ms_per_step"], b["step_ms"]["median"], b["step_ms"]["p10"], b["step_ms"]["p90"], b["learner_steps_per_sec"], b["value"] / 1e3, r["frac"],
        r["avg_launch_ms"], r["frac_of_measured_copy_peak"], cb["value"], cb["linearity_check"]["ratio_to_linear_extrapolation"], b["speedup_vs_cpu_baseline"], cb["all_cores"]["value"])
    mt += "| — without acting (`--no-acting`) / without stored q-values (`--no-policy-outputs`), same box | %.2f / %.2f | | | | |\n" % (na["ms_per_step"], npo["ms_per_step"])
    mt += "| — the tree before `k_gemm3_mid` on ANOTHER box of the pool (`profiles/r06_bench_before_mid_tile_final*.json`): headline / without acting | %.2f / %.2f (acting %.2f ms; final tree: %.2f) | | | | |\n" % (
        old["ms_per_step"], old_na["ms_per_step"], old["ms_per_step"] - old_na["ms_per_step"], b["ms_per_step"] - na["ms_per_step"])
    mt += row("DQN + uniform replay B=256 T=1 (BASELINE `configs[1]`; `other_configs.dqn_uniform`, learner step from a captured graph)", oc["dqn_uniform"]) + "\n"
    mt += row("Rainbow-IQN B=512 T=1, PER cap 2²⁰ (BASELINE `configs[2]`; `other_configs.rainbow_iqn`)", oc["rainbow_iqn"])
    s = re.sub(r"\| config \(1× MI355X, f32 results\) \|.*?\n\| Rainbow-IQN B=512 T=1[^\n]*", lambda m: mt, s, flags=re.S)
    rs = b["roofline_step"]
    whole = ("Whole step: `roofline_step` **%.3f** (%.1f ms of roofline time — every kernel's max(bytes / 8 TB/s, flop / its pipe's dense peak) — in %.2f ms; "
             "librltime_hip kernels %.1f ms measured).  Where the step's time sits: §3's table; by family: split-bf16 GEMMs ≈ 48 ms at 0.47–0.49 of bf16 / 6 "
             "(§3.2: the clock), conv forward 10.0 + backward 6.8 at 0.27–0.52, LSTM 4.8, HBM-bound glue 3.3 at 0.62–0.69, gathers 0.75, acting ≈ %.1f (%.2f vs %.2f "
             "without; 8.8 before `k_gemm3_mid`).  The pool's boxes read one tree within ±1.5 %% (this session's box is the slower kind: its `--no-acting` line is 78.0 against "
             "76.7 on the box of the earlier session); the same-box A/B of the mid tile: 84.61 against 85.29 ms (`profiles/r06_gemm3_mid_tile.jsonl`).  GPU suite in the same "
             "session: 415 passed / 4 skipped, `smoke()` ok; on the last commit of the round (one more RCCL test, the transposed `k_act_embed`): 416 passed / 4 skipped, `smoke()` ok, 84.46 ms per step on that session's box." % (rs["frac"], rs["roofline_ms"], b["ms_per_step"], rs["librltime_hip_measured_ms"], b["ms_per_step"] - na["ms_per_step"], b["ms_per_step"], na["ms_per_step"]))
    s = re.sub(r"Whole step: `roofline_step` \*\*.*?`smoke\(\)` ok\.", lambda m: whole, s, flags=re.S)
    open("DESIGN.md", "w").write(s)


if __name__ == "__main__":
    main()

"""The numbers DESIGN.md section 0 / 6 quote, as markdown, from one evidence set of tools/round_evidence.sh.
usage: python tools/design_numbers.py <dir> <tag>       e.g. python tools/design_numbers.py profiles r5"""
import json
import re
import sys


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main(d, tag):
    f = lambda name: f"{d}/{tag}_{name}"   # noqa: E731
    b = last_json(f("bench_default_run.json"))
    c, r, cb, rl = b["config"], b["roofline"], b["cpu_baseline"], b["real_loop"]
    t = r["temporal_self_attention"]
    clk = c["gpu_clock"]
    print("## headline")
    print(f"step {b['ms_per_step']:.2f} ms = {b['value']:.2f} samples/s @ {clk['sclk_mhz_median']:.0f} MHz {clk.get('power_w_median', 0):.0f} W; "
          f"step_frac {c['step_frac_of_mfma_peak']:.4f} ({c['step_tflops_per_gpu']:.0f} TF/s)")
    print(f"roofline.frac {r['frac']:.4f} ({r['achieved']:.1f} TF/s over {r['kernel_ms_per_step']:.2f} ms, {r['launches']} launches; events {r['kernel_ms_per_step_events']:.2f} ms); "
          f"traffic {r['traffic'] / 1e6:.1f} MB/launch vs algorithmic {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB = {r['traffic'] / r['algorithmic_bytes_per_launch']:.2f}x")
    print(f"TSA op {t['op_frac_of_mfma_peak']:.4f} (fwd {t['fwd_frac_of_mfma_peak']:.4f}, bwd {t['bwd_frac_of_mfma_peak']:.4f}), {t['ms_per_step']:.2f} ms/step")
    print(f"real_loop {rl['ms_per_step']:.2f} ms = {rl['value']:.2f} samples/s (conditioners alone {rl['conditioners_ms_alone']:.2f} ms)")
    pf, pc = cb["parity_full_model"], cb["c2"]["parity_c2"]
    print(f"cpu c1' {cb['seconds']:.1f} s/step ({cb['cores']} cores; loss rel {pf['loss_rel_err']:.1e}, pred {pf['pred_rel_l2']:.2e}); "
          f"c2 {cb['c2']['seconds']:.1f} s (loss rel {pc['loss_rel_err']:.1e}, pred {pc['pred_rel_l2']:.2e})")
    print("\n## results table\n| configuration | ms / optimizer step | samples/s | `roofline.frac` | step frac | loss after the run |\n|---|---|---|---|---|---|")
    rows = [("c2: 14 x 512x320, fp16 (headline)", b)]
    for name, label in (("bf16", "c2, bf16"), ("c5", "c5: c2 + LoRA rank 64, bf16"), ("c4", "c4: 25 x 1024x576, grad-accum 2, fp16")):
        try:
            rows.append((label, last_json(f(f"bench_{name}.json"))))
        except OSError:
            pass
    for label, x in rows:
        xc, xr = x["config"], x.get("roofline") or {}
        num = lambda v, spec: format(v, spec) if isinstance(v, (int, float)) else "-"   # noqa: E731
        print(f"| {label} | {x['ms_per_step']:.2f} | {x['value']:.3f} | {num(xr.get('frac'), '.3f')} | {num(xc.get('step_frac_of_mfma_peak'), '.3f')} | "
              f"{num(xc.get('loss'), '.4f')} ({num(xc.get('opt_steps'), '.0f')} steps) |")
    print("\n## TSA levels\n| rows x channels | blocks | fwd ms (frac) | bwd ms (frac) |\n|---|---|---|---|")
    for lv in t["levels"]:
        print(f"| {lv['rows']} x {lv['channels']} | {lv['fwd']['blocks']} | {lv['fwd']['ms']:.3f} ({lv['fwd']['frac_of_mfma_peak']:.3f}) | {lv['bwd']['ms']:.3f} ({lv['bwd']['frac_of_mfma_peak']:.3f}) |")
    print("\n## categories")
    print(open(f("step_categories.txt")).read().rstrip())
    print("\n## traffic")
    print(open(f("pmc_traffic.txt")).read().rstrip())
    print("\n## mfma busy")
    for line in open(f("pmc_mfma_in_step.txt")):
        if re.match(r"(gemm_|attn_|tsa_|tattn_)", line):
            print(line.rstrip())
    print("\n## A/B (ms added when the path is switched off)")
    for line in open(f("ab_c2.txt")):
        if line.startswith("  "):
            print(line.rstrip())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

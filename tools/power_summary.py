"""Summarise the traces of tools/gpu_power_trace.sh (power_*.csv of tools/power_sampler + the bench lines + the GRBM pass)
into the text committed as profiles/rNN_power_trace.txt.  Usage: python tools/power_summary.py gpurun_out"""
import csv
import json
import os
import sys


def load(path):
    rows, head = [], []
    if not os.path.exists(path):
        return head, rows
    with open(path) as fh:
        lines = [l for l in fh if l.strip()]
    head = [l.strip() for l in lines if l.startswith("#")]
    body = [l for l in lines if not l.startswith("#")]
    for r in csv.DictReader(body):
        try:
            rows.append({k: float(v) for k, v in r.items()})
        except (TypeError, ValueError):
            continue
    return head, rows


def describe(name, rows, busy_w=400.0):
    if not rows:
        print("%s: no samples" % name)
        return
    busy = [r for r in rows if r["socket_power_W"] >= busy_w]
    idle = [r for r in rows if r["socket_power_W"] < busy_w]
    print("%s: %d samples over %.1f s (%.1f Hz); %d under load (socket power >= %.0f W)" % (
        name, len(rows), rows[-1]["t_s"] - rows[0]["t_s"], (len(rows) - 1) / max(1e-9, rows[-1]["t_s"] - rows[0]["t_s"]), len(busy), busy_w))
    if idle:
        print("   idle   : %.0f W mean, gfxclk (mean over XCDs) %.0f MHz mean" % (
            sum(r["socket_power_W"] for r in idle) / len(idle), sum(r["gfxclk_mean_MHz"] for r in idle) / len(idle)))
    if len(busy) >= 4:
        # drop the first / last busy sample (ramp)
        b = busy[1:-1]
        pw = sorted(r["socket_power_W"] for r in b)
        ck = sorted(r["gfxclk_mean_MHz"] for r in b)
        print("   loaded : socket power %.0f W mean (min %.0f, median %.0f, max %.0f); shader clock, mean over the 8 XCDs: %.0f MHz mean (min %.0f, median %.0f, max %.0f); lowest XCD %.0f, highest XCD %.0f MHz" % (
            sum(pw) / len(pw), pw[0], pw[len(pw) // 2], pw[-1], sum(ck) / len(ck), ck[0], ck[len(ck) // 2], ck[-1],
            min(r["gfxclk_min_MHz"] for r in b), max(r["gfxclk_max_MHz"] for r in b)))
        a0, a1 = b[0], b[-1]
        dacc = a1["accumulation_counter"] - a0["accumulation_counter"]
        if dacc > 0:
            print("   residency over the loaded interval (firmware counters, share of accumulation cycles): power limiter (PPT) %.1f %%, PROCHOT %.1f %%, socket thermal %.1f %%, VR thermal %.1f %%, HBM thermal %.1f %%" % tuple(
                100.0 * (a1[k] - a0[k]) / dacc for k in ("ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc")))
        de = a1["energy_acc"] - a0["energy_acc"]
        dt = a1["t_s"] - a0["t_s"]
        if de > 0 and dt > 0:
            print("   energy accumulator: %.0f W over the interval (15.259 uJ units)" % (de * 15.259e-6 / dt))
        print("   hotspot temperature up to %.0f C; GFX activity %.0f %% mean" % (max(r["temp_hotspot_C"] for r in b), sum(r["gfx_activity_pct"] for r in b) / len(b)))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    for tag, label in (("f16", "(a) bench.py --steps 400, fp16 job (sweep_kernel<1> ~80 % of the time)"),
                       ("i8", "(b) bench.py --workload synthetic-u8 --images 192 --steps 100 (sweep_i8_kernel<1>)"),
                       ("ubench", "(c) tools/ubench_clock: short loops, then 10 s of the MFMA-only loop, then 6 s of the VALU-only loop")):
        head, rows = load(os.path.join(out, "power_%s.csv" % tag))
        for h in head:
            print(h)
        describe(label, rows)
        print()
    for f in ("bench_400.json", "bench_i8_100.json"):
        try:
            d = json.loads([l for l in open(os.path.join(out, f)) if l.startswith("{")][0])
            rf = d.get("roofline", {})
            print("%s: %d steps, %.2f ms per step, sweep 1 %.2f ms avg per launch = %.3f of peak (%s)" % (
                f, d["steps"], d["ms_per_step"], rf.get("avg_launch_ms", 0), rf.get("frac", 0), rf.get("unit", "")))
        except Exception as e:  # noqa: BLE001
            print("%s: not readable (%s)" % (f, e))
    try:
        g = json.load(open(os.path.join(out, "pmc_grbm.json")))
        for k, v in g.items():
            if "GRBM_GUI_ACTIVE" in v:
                print("%s: GRBM_GUI_ACTIVE %.4g cycles per launch (divide by the launch duration of the kernel-trace pass for the mean busy clock)" % (
                    k, v["GRBM_GUI_ACTIVE"]["per_launch_mean"]))
    except Exception as e:  # noqa: BLE001
        print("pmc_grbm.json: not readable (%s)" % e)
    try:
        print(open(os.path.join(out, "ubench_clock.txt")).read()[-1800:])
    except OSError:
        pass


if __name__ == "__main__":
    main()

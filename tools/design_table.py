"""DESIGN.md section 8 from the stored bench lines: one row per configuration, a "settled" and a "first 20 after idle"
column, one figure per cell.   python tools/design_table.py [tag]   (reads profiles/<tag>_bench_*.json)"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"


def line(name):
    p = os.path.join(ROOT, "profiles", f"{tag}_bench_{name}.json")
    if not os.path.exists(p):
        return None
    txt = [l for l in open(p).read().splitlines() if l.startswith("{")]
    return json.loads(txt[-1]) if txt else None


rows = [("C2", "C2 1920×1080×3 σ20 psz 8, FLT1 temporal (headline)"),
        ("C2_match_order_block", "C2, `NLK_MATCH_ORDER=block` (opt-in, not the reference's summation order)"),
        ("C3", "C3 3840×2160×3 σ40 psz 12"),
        ("C1", "C1 256×256×1 σ20"),
        ("C1L", "C1L 1920×1080×1 σ20 (gray 1080p)"),
        ("C5", "C5 1080p chain flt1 → flt2 → smo1, resident (3 calls per step)"),
        ("C2_deterministic", "C2 deterministic aggregation"),
        ("C3_deterministic", "C3 deterministic aggregation"),
        ("C2_force_strips_c", "C2, the N > 1 step at N = 1 (`--force-strips`), from C"),
        ("C2_force_strips_c_graph", "the same, replayed HIP graph"),
        ("C2_force_strips_py", "the same, Python driver"),
        ("F1", "F1 TV-L1 flow 1080p (fscale 0)"),
        ("S1", "S1 one frame of the pipeline recursion, resident")]
print("| config | ms / step, settled clocks | first 20 steps after idle | Mpix/s (settled) | dominant launch ms | `frac` / all terms | CPU port Mpix/s (×) | ΔPSNR dB (max-abs) |")
print("|---|---|---|---|---|---|---|---|")
for key, label in rows:
    d = line(key)
    if d is None:
        continue
    r = d.get("roofline") or {}
    cb = d.get("cpu_baseline") or {}
    f20 = d.get("ms_per_step_first_20_unsettled")
    frac = f"{r.get('frac'):.3f}" if r.get("frac") is not None else "—"
    if r.get("frac_all_survey_terms") is not None:
        frac += f" / {r['frac_all_survey_terms']:.3f}"
    cpu = f"{cb['value']:.2f} ({d['speedup_vs_cpu']:.0f}×)" if cb.get("value") and d.get("speedup_vs_cpu") else "—"
    dps = (f"{d['psnr_delta_db']:.3f} ({d['max_abs_vs_cpu']:.1e})" if d.get("psnr_delta_db") is not None else "—")
    lm = r.get("launch_ms")
    print(f"| {label} (`{tag}_bench_{key}.json`) | **{d['ms_per_step']:.4g}** | {f20 if f20 is not None else '—'} | {d['value']:.0f} | "
          f"{lm if lm is not None else '—'} | {frac} | {cpu} | {dps} |")
    if d.get("first_frame_ms"):
        print(f"| … first (spatial) frame of the same size | {d['first_frame_ms']:.4g} | — | {d['value'] * d['ms_per_step'] / d['first_frame_ms']:.0f} | — | — | — | — |")
    if d.get("api_wall_ms") and key in ("C2",):
        print(f"| … drop-in API on pageable host images (PCIe included; never `value`) | {d['api_wall_ms']:.4g} | — | {d['value'] * d['ms_per_step'] / d['api_wall_ms']:.0f} | — | — | — | — |")

"""Prints the per-leg figures of a bench.py JSON line read from stdin (ms per step, fill / traceback, fraction)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    def show(name, o):
        r = o.get("roofline", {})
        k = r.get("ms_per_step_by_kernel", {})
        print(f"{name:14s} {o.get('value', 0):9.1f} {o.get('unit', '')}  {o.get('ms_per_step', 0):9.3f} ms/step  "
              f"pack {k.get('pack', 0):.3f} fill {k.get('fill', 0):.3f} tb {k.get('traceback', 0):.3f}  "
              f"frac {r.get('frac', 0):.3f} ({r.get('kernel', '')})")
    show("headline", d)
    for name, leg in d.get("legs", {}).items():
        show(name, leg)
    for name, leg in d.get("minimizers", {}).items():
        if isinstance(leg, dict):
            print(f"min {name:22s} {leg.get('ms', 0):.3f} ms  frac {leg.get('roofline', {}).get('frac', 0):.4f}")

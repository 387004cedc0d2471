"""tools/bench_brief.py <bench json>: the handful of numbers one looks at first."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
keys = ("value", "ms_per_step", "ms_per_step_unpipelined", "value_pcie", "ms_per_step_pcie", "value_single_segment", "value_single_segment_pcie",
        "lone_segment_ms", "checked_max_abs", "outputs_finite")
print({k: d.get(k) for k in keys})
for k in d.get("kernels", []):
    print(f"  {k['kernel'][:36]:36s} x{k['launches_per_step']:<3d} {k['launch_ms']:8.4f} ms (alone {k['launch_ms_alone']:8.4f})  frac {k.get('frac')} issued {k.get('frac_issued')}")

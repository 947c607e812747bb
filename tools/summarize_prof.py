"""Condenses rocprofv3 CSV output into the small summaries committed under profiles/.

    python tools/summarize_prof.py stats <dir> [last=N]    per-kernel duration from the kernel trace
    python tools/summarize_prof.py pmc   <dir> [last=N]    per-kernel average of the collected counter

`last=N` keeps only the last N launches of every kernel (the bench's warm-up + timed steps), which drops the
set-up launches (teacher rendering of the targets uses 32768-ray super-chunks)."""
import collections
import csv
import glob
import os
import sys


def _ours(name):
    return any(k in name for k in ("march_", "shade_", "composite_kernel", "reduce_replicas", "pack_", "bin_", "tile_", "adam_kernel",
                                   "wslab_"))


def stats(d, last):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows = []
    steps = min((len(v) for k, v in per.items() if "adam_kernel" in k or "shade_backward" in k), default=0)
    for k, v in per.items():
        v.sort()
        per_step = len(v) // steps if steps and len(v) % steps == 0 else 1   # e.g. the scatter kernels run twice per step
        dur = [x[1] for x in (v[-last * per_step:] if last and _ours(k) else v)]
        rows.append((sum(dur), k, len(dur), sum(dur) / len(dur), min(dur), max(dur)))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    out = ["name,calls,avg_us,min_us,max_us,total_ms,percent"]
    for t, k, n, avg, mn, mx in rows[:16]:
        out.append(f"\"{k[:110]}\",{n},{avg/1e3:.1f},{mn/1e3:.1f},{mx/1e3:.1f},{t/1e6:.3f},{100*t/tot:.2f}")
    return "\n".join(out)


def pmc(d, last):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if _ours(r["Kernel_Name"]):
            acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = ["kernel,counter,launches,avg_value"]
    for k, cs in acc.items():
        for c, v in cs.items():
            v.sort()
            vals = [x[1] for x in (v[-last:] if last else v)]
            out.append(f"\"{k}\",{c},{len(vals)},{sum(vals)/len(vals):.1f}")
    return "\n".join(out)


if __name__ == "__main__":
    kind, d = sys.argv[1], sys.argv[2]
    last = 0
    for a in sys.argv[3:]:
        if a.startswith("last="):
            last = int(a[5:])
    print(stats(d, last) if kind == "stats" else pmc(d, last))

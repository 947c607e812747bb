"""Mean duration of every kernel over the hipGraph replays of a `rocprofv3 --kernel-trace` run of bench.py (replays = the
steps whose Adam launches are 0.65-0.9 ms apart)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70]) for r in csv.DictReader(open(f))]
rows.sort()
adam = [r for r in rows if 'adam_kernel' in r[2]]
acc = collections.defaultdict(list)
steps = []
for i in range(len(adam) - 1):
    d = (adam[i + 1][0] - adam[i][0]) / 1e3
    if 650 < d < 900:
        steps.append(d)
        for s, e, n in rows:
            if adam[i][1] <= s <= adam[i + 1][1]:
                acc[n.replace('(anonymous namespace)::', '')[:40]].append((e - s) / 1e3)
print("replays", len(steps), "mean step", sum(steps) / max(len(steps), 1))
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:42s} n/step {len(v)/max(len(steps),1):4.1f}  mean {sum(v)/len(v):7.1f} us")

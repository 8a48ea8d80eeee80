"""Timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py: per step, how long is the GPU busy (union over the streams), how
long is each stream busy, and where are the idle gaps?

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-parity --no-roofline
    python tools/trace_gaps.py out/**/*_kernel_trace.csv > profiles/r04_timeline_gaps.txt
"""
import collections
import csv
import sys


def main():
    rows = []
    for path in sys.argv[1:]:
        with open(path) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    # steps are delimited by the fused Adam launch of the big parameter run (the longest adam_kernel of every step)
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2] and r[1] - r[0] > 50_000]
    if len(adam) < 3:
        print("not enough steps in the trace", len(adam))
        return
    print(f"# {len(rows)} kernel records, {len(adam)} steps (delimited by the large adam_kernel launch)")
    print("# step: wall us | union busy us | idle us | gaps > 3 us: count, total us | busy per queue us")
    tot_gap = collections.Counter()
    for k in range(len(adam) - 3, len(adam) - 1):           # the last two complete steps
        a, b = adam[k] + 1, adam[k + 1] + 1
        seg = rows[a:b]
        t0, t1 = rows[adam[k]][1], seg[-1][1]
        busy, cur_s, cur_e = 0, None, None
        gaps = []
        prev_name = ""
        for s, e, name, q, st in sorted(seg):
            if cur_e is None:
                cur_s, cur_e = s, e
                prev_name = name
                if s - t0 > 3000:
                    gaps.append((s - t0, "step start", name))
                continue
            if s > cur_e:
                busy += cur_e - cur_s
                if s - cur_e > 3000:
                    gaps.append((s - cur_e, prev_name, name))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
            prev_name = name if e >= cur_e else prev_name
        busy += cur_e - cur_s
        perq = collections.Counter()
        for s, e, name, q, st in seg:
            perq[q] += e - s
        wall = t1 - t0
        print(f"step {k}: wall {wall / 1e3:9.1f} | busy {busy / 1e3:9.1f} | idle {(wall - busy) / 1e3:8.1f} | {len(gaps):3d} gaps, "
              f"{sum(g[0] for g in gaps) / 1e3:7.1f} us | " + ", ".join(f"q{q}: {v / 1e3:.0f}" for q, v in sorted(perq.items())))
        for g, before, after in sorted(gaps, reverse=True)[:12]:
            print(f"      gap {g / 1e3:6.1f} us  after {before[27:80]:55s} before {after[27:80]}")
    # launches per step and sub-16-us launches
    a, b = adam[-2] + 1, adam[-1] + 1
    seg = rows[a:b]
    small = [r for r in seg if r[1] - r[0] < 16000]
    print(f"# last step: {len(seg)} launches, {len(small)} of them under 16 us ({sum(r[1] - r[0] for r in small) / 1e3:.0f} us in total)")


if __name__ == "__main__":
    main()

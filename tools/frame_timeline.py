#!/usr/bin/env python3
"""One frame of a traced loop as a timeline: python tools/frame_timeline.py <kernel_trace.csv> <memory_copy_trace.csv> [anchor kernel] [which occurrence]
Prints every kernel and copy between one launch of the anchor kernel (default k_stereo_prune, the last kernel of the stereo constructor) and the next."""
import csv
import sys


def main():
    kt, mt = sys.argv[1], sys.argv[2]
    anchor = sys.argv[3] if len(sys.argv) > 3 else "k_stereo_prune"
    which = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    ev = []
    for r in csv.DictReader(open(kt)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, "stream %s" % r.get("Stream_Id", "?")))
    for r in csv.DictReader(open(mt)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "")).replace("MEMORY_COPY_", ""), "stream %s" % r.get("Stream_Id", "?")))
    ev.sort()
    idx = [i for i, e in enumerate(ev) if e[2].startswith(anchor)]
    a, b = idx[which], idx[which + 1]
    t0 = ev[a][0]
    print(f"start since the anchor ({anchor} #{which}), duration, kernel / copy, HIP stream")
    for s, e, n, st in ev[a:b + 1]:
        print(f"{(s - t0) / 1e3:9.1f} us  + {(e - s) / 1e3:6.1f} us  {n:34s} {st}")


if __name__ == "__main__":
    main()

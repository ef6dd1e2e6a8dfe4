#!/usr/bin/env python3
"""Per-frame view of a rocprofv3 kernel trace of bench.py (…_kernel_trace.csv): a frame starts at df_pack_bounds_kernel (set_transforms);
for every frame the start / end of each kernel relative to it, with the queue it ran on (the look-ahead builds run on the warp field's
side stream beside the sweep).  usage: tools/frame_trace.py trace_kernel_trace.csv [first_frame last_frame]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 60)
frames, cur = [], None
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("df_pack_bounds"):
        cur = []; frames.append(cur)
    if cur is not None and k.startswith("df_"):
        cur.append((k.split("<")[0].replace("df_", "").replace("_kernel", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
# the TIMED frames of `bench.py --steps 20 --warmup 5` are frames 9..28 here (frame 0 = WarpField.init's pack, 1-3 priming, 4-8 warm-up):
# mean duration per kernel over them -- what kernel_ms.integrate_warped (HIP events around the whole call: pyramid + verdict pass +
# urgent build + plan + sweep + the side stream's fork) is to be compared with
t_lo, t_hi = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (9, 28)
acc = {}
for f in frames[t_lo:t_hi + 1]:
    for k, s_, e_, q in f:
        acc.setdefault((k, q), []).append((e_ - s_) / 1e3)
print("timed frames %d..%d, mean kernel duration (us) [queue]: " % (t_lo, t_hi) +
      ", ".join("%s[q%s] %.1f" % (k, q, sum(v) / (t_hi - t_lo + 1)) for (k, q), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))))
print("%d frames; per frame: kernel[queue] start-end (us after the frame's first kernel); period = start of the next frame" % len(frames))
for i, f in enumerate(frames[:-1]):
    if i < lo or i > hi or not f: continue
    t0 = f[0][1]
    print("frame %3d period %7.1f | " % (i, (frames[i + 1][0][1] - t0) / 1e3) + " | ".join("%s[q%s] %.0f-%.0f" % (k[:14], q, (s - t0) / 1e3, (e - t0) / 1e3) for k, s, e, q in f))

#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel (per launch).

usage: pmc_summary.py [root=gpurun_out/pmc] [--json profiles/pmc_latest.json --config 512] [--tag "round 1"]

HBM bytes per launch follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced reads, so fetch bytes = 2 * FETCH_SIZE * 1024
(cross-checked against TCC_EA0_RDREQ * 128, collected in its own pass); write bytes = WRITE_SIZE * 1024.
"""
import argparse, sys, collections, csv, glob, hashlib, json, os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha(kernel):
    """sha256 of the sources a kernel is built from -- its .hip and the device headers: bench.py only reports counters taken from the
    build it runs (dynamicfusion_amd.build.kernel_source_sha is the same function)"""
    sys.path.insert(0, REPO)
    from dynamicfusion_amd import build as B
    return B.kernel_source_sha(kernel)

ap = argparse.ArgumentParser()
ap.add_argument("root", nargs="?", default="gpurun_out/pmc")
ap.add_argument("--json", default=None)
ap.add_argument("--config", default="512")
ap.add_argument("--tag", default="round 1")
ap.add_argument("--last", type=int, default=0, help="use only the last N launches of every kernel in each pass (the ones after the warm-up)")
args = ap.parse_args()

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(args.root, "*", "*counter_collection.csv"))):
    per_file = collections.defaultdict(lambda: collections.defaultdict(list))
    recs = list(csv.DictReader(open(f)))
    if recs and "Dispatch_Id" in recs[0]:
        recs.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in recs:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("df_"):
            continue
        per_file[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in per_file:
        for c, v in per_file[k].items():
            rows[k][c].extend(v[-args.last:] if args.last else v)

workload = {}
try:
    workload = json.load(open(os.path.join(REPO, "gpurun_out", "pmc_workload.json"))).get(args.config, {})
except Exception:
    pass
out = {}
for k in sorted(rows):
    print(k)
    mean = {c: sum(v) / len(v) for c, v in rows[k].items()}
    for c in sorted(mean):
        print("    %-22s %.6g" % (c, mean[c]))
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        fetch, write = 2.0 * mean["FETCH_SIZE"] * 1024, mean["WRITE_SIZE"] * 1024
        rd = mean.get("TCC_EA0_RDREQ_sum", float("nan")) * 128
        print("    %-22s %.4g GB (fetch %.4g + write %.4g; RDREQ*128 = %.4g)" %
              ("hbm_bytes/launch", (fetch + write) / 1e9, fetch / 1e9, write / 1e9, rd / 1e9))
        out[k.split("<")[0]] = {"kernel": k, "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                                "tcc_ea0_rdreq_x128": rd, "source_sha256": source_sha(k),
                                "how": "rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, %s" % args.tag,
                                # every counter collected for the kernel (mean per launch, separate --pmc passes): bench.py's
                                # roofline.compute reads SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES from here
                                "counters": {c: mean[c] for c in sorted(mean)}}
        if k.startswith("df_warp_rows_pipe_kernel") and workload.get("n_swept_per_launch"):
            out[k.split("<")[0]]["n_swept_per_launch"] = workload["n_swept_per_launch"]
    if "SQ_BUSY_CYCLES" in mean and "SQ_ACTIVE_INST_VALU" in mean and "SQ_WAVES" in mean:
        print("    %-22s %.4g per wave" % ("VALU instr", mean.get("SQ_INSTS_VALU", 0) / max(mean["SQ_WAVES"], 1)))

if args.json:
    doc = {}
    if os.path.exists(args.json):
        try:
            doc = json.load(open(args.json))
        except Exception:
            doc = {}
    doc[args.config] = out
    json.dump(doc, open(args.json, "w"), indent=1)
    print("wrote", args.json)

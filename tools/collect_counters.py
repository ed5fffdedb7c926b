"""Regenerates profiles/hbm_traffic.json and the per-workload counter summaries it is made of.

For every (workload, NN kernel) bench.py reports a roofline for, run tools/one_align.py under
rocprofv3 in SEPARATE passes -- kernel trace + stats, then one --pmc pass per counter group
(FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; MI355X_MICROARCH.md, rocprofv3 PMC slots)
-- and average every counter over the dispatches of the NN kernel (first sweeps in the proportion
they occur in an alignment).  HBM bytes per launch: reads = 128 x TCC_EA0_RDREQ_128B + 64 x TCC_EA0_RDREQ_64B +
32 x TCC_EA0_RDREQ_32B (the L2's memory-side read requests by size), writes = WRITE_SIZE x 1024 (= 64 x WRREQ_64B +
32 x the rest).  Calibrated on known byte counts in this kernel's access shapes (tools/microbench_traffic.hip,
profiles/r03_traffic_calibration.txt): every read request on gfx950 is a 128-byte line -- a 64-byte gather by 4 lanes
fetches 128, a 4-byte cell_start look-up fetches 128 -- which FETCH_SIZE tallies at 64 bytes, so 2 x FETCH_SIZE equals
the request-size sum (kept as a cross-check); WRITE_SIZE is exact (32-byte sectors: a scattered 4- or 8-byte store
costs 32 bytes, two adjacent 16-byte stores 33.6).

Run on the GPU box:   python3 tools/collect_counters.py [--only key,key] [--tag r02]
Outputs (tracked):    profiles/<tag>_<workload>_<mode>_counters.csv, profiles/<tag>_<workload>_<mode>_kernel_stats.csv,
                      profiles/hbm_traffic.json
Every pass's rocprofv3 log is kept under gpurun_out/counters/."""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOBS = {  # key -> (one_align arguments, substring naming the NN kernel)
    "kinect640x480_30pct:grid": (["--workload", "kinect640x480_30pct", "--nn-mode", "grid"], "nn_grid_kernel"),
    "kinect640x480_30pct:filtered": (["--workload", "kinect640x480_30pct", "--nn-mode", "filtered", "--iters", "4"], "nn_filtered_kernel"),
    "kinect640x480_30pct:pruned": (["--workload", "kinect640x480_30pct", "--nn-mode", "pruned"], "nn_pruned_kernel"),
    "kinect640x480_dense:grid": (["--workload", "kinect640x480_dense", "--nn-mode", "grid"], "nn_grid_kernel"),
    "dense1m:grid": (["--workload", "dense1m", "--nn-mode", "grid", "--iters", "50", "--reps", "1"], "nn_grid_kernel"),
    "frame_batch8:grid": (["--batch", "8"], "nn_grid_batch_kernel"),
    "frame_batch16:grid": (["--batch", "16"], "nn_grid_batch_kernel"),  # the launch shape of bench.py's frame_batch block (groups of 16)
    "kinect_v2_512x424:grid": (["--workload", "kinect_v2_512x424", "--nn-mode", "grid"], "nn_grid_kernel"),
}
RD = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], RD, ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_SALU"], ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"]]


def run(cmd, log, timeout):
    with open(log, "w") as f:
        try:
            r = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=timeout, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp"))
            return r.returncode
        except subprocess.TimeoutExpired:
            f.write("\nTIMEOUT\n")
            return 124


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--timeout", type=int, default=240)
    ap.add_argument("--mirror", default=os.path.join(ROOT, "gpurun_out", "profiles_new"),
                    help="second copy of every output (gpurun merges only gpurun_out/ back: copy it into profiles/)")
    a = ap.parse_args()
    keys = [k for k in JOBS if not a.only or k in a.only.split(",")]
    logdir = os.path.join(ROOT, "gpurun_out", "counters")
    os.makedirs(logdir, exist_ok=True)
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    table = json.load(open(tfile)) if os.path.exists(tfile) else {}
    table = {k: v for k, v in table.items() if isinstance(v, dict)}  # drop round-1 scalar entries
    for key in keys:
        oargs, kname = JOBS[key]
        slug = key.replace(":", "_")
        prog = ["python3", os.path.join(ROOT, "tools", "one_align.py")] + oargs
        # pass 0: kernel trace + stats (durations)
        d = f"/tmp/cc_{slug}_trace"
        shutil.rmtree(d, ignore_errors=True)
        rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "p", "--"] + prog,
                 os.path.join(logdir, f"{slug}_trace.log"), a.timeout)
        print(key, "trace rc", rc, flush=True)
        if rc != 0:
            continue
        dur = None
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            shutil.copy(f, os.path.join(ROOT, "profiles", f"{a.tag}_{slug}_kernel_stats.csv"))
            rows = [r for r in csv.DictReader(open(f)) if kname in r["Name"]]
            calls = sum(int(r["Calls"]) for r in rows)
            if calls:
                dur = sum(float(r["TotalDurationNs"]) for r in rows) / calls
        acc = collections.defaultdict(lambda: [0, 0.0])  # (kernel, counter) -> [dispatches, sum]
        ok = True
        for i, grp in enumerate(PASSES):
            d = f"/tmp/cc_{slug}_pmc{i}"
            shutil.rmtree(d, ignore_errors=True)
            rc = run(["rocprofv3", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + prog,
                     os.path.join(logdir, f"{slug}_pmc{i}.log"), a.timeout)
            print(key, "pmc", grp, "rc", rc, flush=True)
            if rc != 0:
                ok = ok and i >= 3  # the traffic passes are mandatory, the instruction mix is not
                continue
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"].split("(")[0]
                    e = acc[(name, r["Counter_Name"])]
                    e[0] += 1
                    e[1] += float(r["Counter_Value"])
        with open(os.path.join(ROOT, "profiles", f"{a.tag}_{slug}_counters.csv"), "w") as f:
            f.write("kernel,counter,dispatches,avg_per_dispatch\n")
            for (k, c), (n, v) in sorted(acc.items()):
                f.write(f"\"{k}\",{c},{n},{v / n:.3f}\n")
        if not ok:
            continue

        def avg(counter):
            n = sum(v[0] for (k, c), v in acc.items() if c == counter and kname in k)
            s = sum(v[1] for (k, c), v in acc.items() if c == counter and kname in k)
            return s / n if n else None

        fetch_kb, write_kb = avg("FETCH_SIZE"), avg("WRITE_SIZE")
        n32, n64, n128 = avg("TCC_EA0_RDREQ_32B_sum"), avg("TCC_EA0_RDREQ_64B_sum"), avg("TCC_EA0_RDREQ_128B_sum")
        if fetch_kb is None or write_kb is None or n128 is None:
            continue
        read_bytes = 128.0 * n128 + 64.0 * n64 + 32.0 * n32
        table[key] = {
            "kernel": kname, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
            "read_requests": {"32B": n32, "64B": n64, "128B": n128}, "read_bytes_per_launch": read_bytes,
            "read_bytes_over_2x_fetch_size": read_bytes / (2.0 * fetch_kb * 1024.0) if fetch_kb else None,
            "write_bytes_per_launch": write_kb * 1024.0,
            "hbm_bytes_per_launch": read_bytes + write_kb * 1024.0,
            "valu_insts_per_launch": avg("SQ_INSTS_VALU"), "salu_insts_per_launch": avg("SQ_INSTS_SALU"),
            "waves_per_launch": avg("SQ_WAVES"), "vmem_rd_insts_per_launch": avg("SQ_INSTS_VMEM_RD"),
            "vmem_wr_insts_per_launch": avg("SQ_INSTS_VMEM_WR"), "lds_insts_per_launch": avg("SQ_INSTS_LDS"),
            "rocprof_avg_launch_ns": dur,
            "source": f"profiles/{a.tag}_{slug}_counters.csv, profiles/{a.tag}_{slug}_kernel_stats.csv (tools/collect_counters.py)",
        }
        print(key, json.dumps(table[key]), flush=True)
    table["_note"] = ("written by tools/collect_counters.py; hbm_bytes_per_launch = 128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B "
                      "(TCC_EA0 read requests by size) + WRITE_SIZE*1024, averaged over the NN kernel's dispatches, separate "
                      "rocprofv3 --pmc passes; calibration of these counters on known byte counts in the kernel's access shapes: "
                      "profiles/r03_traffic_calibration.txt (tools/pmc_traffic.sh); 2 x FETCH_SIZE is kept as a cross-check")
    json.dump(table, open(tfile, "w"), indent=1, sort_keys=True)
    print("wrote", tfile)
    if a.mirror:
        os.makedirs(a.mirror, exist_ok=True)
        for f in glob.glob(os.path.join(ROOT, "profiles", f"{a.tag}_*")) + [tfile]:
            shutil.copy(f, a.mirror)


if __name__ == "__main__":
    sys.exit(main())

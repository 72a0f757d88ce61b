"""Rewrite the entries of profiles/traffic.json and profiles/limiter.json that bench.py reads, from summarised PMC passes
(profiles/<tag>/pmc_*.json, written by summarise.py), stamping them with the build they were measured on.

    python profiles/refresh_side_data.py <workload> <replicas> <kernel name bench.py reports> <tag> <pmc json> [<pmc json> ...]
Several pmc files (the two kernels of the hybrid tick) are summed.  HBM-side bytes per launch, calibrated against known byte counts
in this engine's access patterns (profiles/ubench/bytes_calib.json): reads = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B
(FETCH_SIZE tallies every request as 64 B: exact for isolated 64-byte requests, half for whole 128-byte lines), writes = WRITE_SIZE
(= distinct 32-byte sectors written x 32, exact in every pattern measured).  Profiles without the request-size pass fall back to
FETCH_SIZE x 2 (an upper bound) and say so."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload, replicas, kernel, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
files = sys.argv[5:]
fetch = write = valu = wait = cyc = 0.0
rd_exact = 0.0
have_sizes = True
build = None
for f in files:
    d = json.load(open(os.path.join(ROOT, "profiles", tag, f)))
    fetch += d["FETCH_SIZE"]["mean_per_launch"] * 1024
    write += d["WRITE_SIZE"]["mean_per_launch"] * 1024
    if "TCC_EA0_RDREQ_128B_sum" in d:
        rd_exact += 32 * d["TCC_EA0_RDREQ_32B_sum"]["mean_per_launch"] + 64 * d["TCC_EA0_RDREQ_64B_sum"]["mean_per_launch"] + 128 * d["TCC_EA0_RDREQ_128B_sum"]["mean_per_launch"]
    else:
        have_sizes = False
    valu += d["SQ_INSTS_VALU"]["mean_per_launch"]
    wait += d["SQ_WAIT_ANY"]["mean_per_launch"]
    cyc += d["SQ_WAVE_CYCLES"]["mean_per_launch"]
    build = d.get("_build", build)
src = " + ".join("profiles/%s/%s" % (tag, f) for f in files)
for name, entry in (("traffic.json", {"workload": workload, "replicas": replicas, "kernel": kernel, "build": (build or "").split("+")[0], "fetch_bytes_raw": fetch, "write_bytes": write,
                                      "read_bytes": rd_exact if have_sizes else 2 * fetch,
                                      "read_bytes_basis": "32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B (calibrated: profiles/ubench/bytes_calib.json)" if have_sizes else "FETCH_SIZE x 2 (upper bound: no request-size pass)",
                                      "hbm_bytes_per_launch": (rd_exact if have_sizes else 2 * fetch) + write, "source": src}),
                    ("limiter.json", {"workload": workload, "replicas": replicas, "kernel": kernel, "build": (build or "").split("+")[0], "valu_insts_per_launch": valu,
                                      "cycles_per_valu_inst": 4.0, "simds": 1024, "clock_hz": 2400000000.0, "wave_wait_frac": round(wait / cyc, 3),
                                      "source": src + " + profiles/ubench/issue_rate.txt"})):
    path = os.path.join(ROOT, "profiles", name)
    doc = json.load(open(path))
    doc["entries"] = [e for e in doc["entries"] if not (e.get("workload") == workload and e.get("replicas") == replicas and e.get("kernel") == kernel)] + [entry]
    json.dump(doc, open(path, "w"), indent=1)
    print(name, json.dumps(entry))

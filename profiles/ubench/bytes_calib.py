"""Join the known byte counts of profiles/ubench/bytes_calib.hip with the per-kernel counter means of its rocprofv3 passes
(gpurun_out/bytes_calib/, written by bytes_calib.sh) -> profiles/ubench/bytes_calib.json: per access pattern
    fetch_factor = known read bytes / FETCH_SIZE bytes      write_factor = known written bytes / WRITE_SIZE bytes
for three notions of "known": useful bytes, distinct 32-byte sectors, distinct 128-byte lines.  The factor to multiply a kernel's
FETCH_SIZE / WRITE_SIZE with is the one whose notion matches what the memory system has to move for that pattern."""
import glob, json, os, re, sys
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "gpurun_out", "bytes_calib")
known = [json.loads(l) for l in open(os.path.join(src, "known.jsonl")) if l.startswith("{")]

def means(pass_name):
    fs = glob.glob(os.path.join(src, pass_name, "*", "*_counter_collection.csv"))
    if not fs:
        return {}
    df = pd.read_csv(max(fs, key=os.path.getmtime))
    out = {}
    for (k, c), g in df.groupby(["Kernel_Name", "Counter_Name"]):
        v = g["Counter_Value"].values
        out[(re.sub(r"^void ", "", k).split("(")[0], c)] = float(v[1:].mean() if len(v) > 1 else v.mean())     # (first launch: cold)
    return out

cnt = {}
for p in ("fetch", "write", "rdreq", "wrreq", "hit"):
    cnt.update(means(p))
rows = []
for k in known:
    name = k["kernel"]
    g = lambda c: cnt.get((name, c))
    fetch_kb, write_kb = g("FETCH_SIZE"), g("WRITE_SIZE")
    r = {"kernel": name, "ms": k["ms"], "read_known": k["read"], "write_known": k["write"],
         "FETCH_SIZE_bytes": None if fetch_kb is None else fetch_kb * 1024, "WRITE_SIZE_bytes": None if write_kb is None else write_kb * 1024,
         "TCC_EA0_RDREQ": g("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_32B": g("TCC_EA0_RDREQ_32B_sum"),
         "TCC_EA0_RDREQ_64B": g("TCC_EA0_RDREQ_64B_sum"), "TCC_EA0_RDREQ_128B": g("TCC_EA0_RDREQ_128B_sum"), "TCC_EA0_ATOMIC": g("TCC_EA0_ATOMIC_sum"),
         "TCC_EA0_WRREQ": g("TCC_EA0_WRREQ_sum"), "TCC_EA0_WRREQ_64B": g("TCC_EA0_WRREQ_64B_sum"),
         "TCC_HIT": g("TCC_HIT_sum"), "TCC_MISS": g("TCC_MISS_sum")}
    if r["TCC_EA0_RDREQ_128B"] is not None:
        r["read_bytes_by_request_size"] = 32 * (r["TCC_EA0_RDREQ_32B"] or 0) + 64 * (r["TCC_EA0_RDREQ_64B"] or 0) + 128 * r["TCC_EA0_RDREQ_128B"]
        if k["read"]["useful"] > 0:
            r["request_size_factor"] = {n: k["read"][n] / max(r["read_bytes_by_request_size"], 1) for n in ("useful", "sectors", "lines")}
    for side, ctr in (("read", "FETCH_SIZE_bytes"), ("write", "WRITE_SIZE_bytes")):
        if r[ctr] and k[side]["useful"] > 0:
            r[side + "_factor"] = {n: k[side][n] / r[ctr] for n in ("useful", "sectors", "lines") if k[side][n] > 0}
    rows.append(r)
out = {"_note": "known bytes / counter bytes per access pattern (rocprofv3 on gfx950, ROCm 7.2; FETCH_SIZE / WRITE_SIZE reported in KiB); "
                "means over launches 2..5 of each kernel; see bytes_calib.hip for the patterns", "patterns": rows}
json.dump(out, open(os.path.join(ROOT, "profiles", "ubench", "bytes_calib.json"), "w"), indent=1)
for r in rows:
    f = lambda d: " ".join("%s %.2f" % kv for kv in (d or {}).items())
    print("%-28s %7.3f ms  FETCH %9.1f MB [%s]  by request size %9.1f MB [%s]  WRITE %9.1f MB [%s]" % (
        r["kernel"], r["ms"], (r["FETCH_SIZE_bytes"] or 0) / 1e6, f(r.get("read_factor")), (r.get("read_bytes_by_request_size") or 0) / 1e6, f(r.get("request_size_factor")),
        (r["WRITE_SIZE_bytes"] or 0) / 1e6, f(r.get("write_factor"))))

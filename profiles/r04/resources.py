"""Register / scratch / occupancy table of the kernels of one source file, from the compiler's own report:
    python profiles/r04/resources.py vds_tick_dense [extra hipcc flags]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(ROOT, "vehicles_dispatch_simulator_amd", "csrc", f + ".hip"),
       "-o", "/tmp/%s_res.o" % f, "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for l in err.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", l)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(.*", "", n).replace("void vds::", "")
    print("%-44s VGPR %3s AGPR %3s SGPR %3s scratch %4s B/lane  waves/SIMD %s  SGPR spills %4s  VGPR spills %4s  LDS %s" % (
        n, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))

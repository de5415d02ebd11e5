"""Developer aid: registers / spills / scratch / LDS / occupancy of every kernel, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/resource_report.py [extra hipcc flags]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "mpc-for-dynamic-locomotion-in-the-mit-cheetah-3_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", "mpcqp_kernels.hip"] + sys.argv[1:]
out = subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
cur = None
rows = {}
for l in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", l)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: +(\d+)", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':90s} VGPR AGPR spillV spillS scratch   LDS occ")
for k, r in rows.items():
    print(f"{k[:90]:90s} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('VGPRs Spill', 0):6d} {r.get('SGPRs Spill', 0):6d} {r.get('ScratchSize', 0):7d} {r.get('LDS Size', 0):5d} {r.get('Occupancy', 0):3d}")

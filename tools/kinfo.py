#!/usr/bin/env python3
"""Register / LDS / scratch footprint and instruction counts of the kernels of one csrc/*.hip file (runs without a GPU).
   python tools/kinfo.py k_corners.hip [regex]   -> one line per kernel whose demangled name matches"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd", "csrc")
EXTRA = {"k_corners.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]}

def main():
    src = sys.argv[1]; pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    extra = EXTRA.get(os.path.basename(src), []) + sys.argv[3:]
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    path = src if os.path.exists(src) else os.path.join(CS, src)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           f"-I{ROOT}/include", f"-I{CS}", "--cuda-device-only", "-S", *extra, path, "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
    names = [l.split(":")[0] for l in lines if re.match(r"^_Z\w+:", l)]
    for sym in names:
        dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        if not pat.search(dem): continue
        i0 = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
        i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end") or ".amdhsa_kernel" in lines[i])
        body = lines[i0:i1]
        n = {k: sum(1 for l in body if re.match(r"^\s+" + k, l)) for k in ("v_", "s_", "ds_", "global_|buffer_|flat_")}
        meta = {}
        k0 = next(i for i, l in enumerate(lines) if l.strip() == ".amdhsa_kernel " + sym)
        for l in lines[k0:k0 + 60]:
            m = re.match(r"^\s+\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)\s+(\d+)", l)
            if m: meta[m.group(1)] = int(m.group(2))
        print(f"{dem[:90]:90s} vgpr {meta.get('next_free_vgpr')} sgpr {meta.get('next_free_sgpr')} lds {meta.get('group_segment_fixed_size')} scratch {meta.get('private_segment_fixed_size')} | static VALU {n['v_']} SALU {n['s_']} LDS {n['ds_']} VMEM {n['global_|buffer_|flat_']}")
    os.unlink(out)
main()

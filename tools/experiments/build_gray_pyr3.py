#!/usr/bin/env python3
"""Builds build_variants/libofk_graypyr3{w,c}.so: the product library with tools/experiments/k_gray_pyr3.hip (gray conversion fused into the
three-level pyramid pass) pasted into k_image.hip and called by ofk_pairs_run for the previous frames, then the next frames (two launches, so
the response kernel still waits for the previous frames only).  w = one-wave workgroups with the product pass's chunking (8192 waves),
c = 256-thread workgroups with 2048 waves as measured in round 2.  Nothing here touches the product sources.
   python tools/experiments/build_gray_pyr3.py && bash tools/experiments/run_variants.sh"""
import os, re, shutil, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd", "csrc")
OUT = os.path.join(ROOT, "build_variants")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", f"-I{ROOT}/include"]

def build(name, one_wave):
    tmp = tempfile.mkdtemp()
    for f in os.listdir(CS):
        if f.endswith((".hip", ".h")): shutil.copy(os.path.join(CS, f), tmp)
    exp = open(os.path.join(ROOT, "tools", "experiments", "k_gray_pyr3.hip")).read()
    exp = exp.replace('getenv("OFK_NO_GRAY_PYR3") != nullptr', 'false').replace('if (const char *e = getenv("OFK_PYR3_CHUNKS")) { const int v = atoi(e); if (v >= 1 && v <= maxchunks) nchunks = v; }   // tuning knob', '')
    exp = exp.replace('if (getenv("OFK_GP3_NQ4")) hipLaunchKernelGGL(k_gray_pyr3<4>, grid, dim3(256), 0, s, A);\n    else hipLaunchKernelGGL', 'hipLaunchKernelGGL')
    if one_wave:
        exp = exp.replace("__global__ __launch_bounds__(256) void k_gray_pyr3", "__global__ __launch_bounds__(64) void k_gray_pyr3")
        exp = exp.replace("const int wid = __builtin_amdgcn_readfirstlane(bx * 4 + (threadIdx.x >> 6));", "const int wid = bx;")
        exp = exp.replace("int nchunks = (2048 + images * nstrips - 1)", "int nchunks = (8192 + images * nstrips - 1)")
        exp = exp.replace("dim3 grid((nstrips * nchunks + 3) / 4, images);", "dim3 grid(nstrips * nchunks, images);")
        exp = exp.replace("hipLaunchKernelGGL(k_gray_pyr3<8>, grid, dim3(256), 0, s, A);", "hipLaunchKernelGGL(k_gray_pyr3<8>, grid, dim3(64), 0, s, A);")
    img = open(os.path.join(tmp, "k_image.hip")).read()
    anchor = "static bool pyr_stream_ok(int h, int w)"
    img = img.replace(anchor, exp + "\n" + anchor)
    open(os.path.join(tmp, "k_image.hip"), "w").write(img)
    hdr = open(os.path.join(tmp, "ofk_internal.h")).read()
    hdr = hdr.replace("bool ofk_launch_pyr3(", "bool ofk_launch_gray_pyr3(hipStream_t s, const uint8_t *bgr0, const uint8_t *bgr1, size_t bgr_stride, uint8_t *pyr0, uint8_t *pyr1,\n"
                      "                          size_t stride, const ofk_levels &lv, int batch, int images);\nbool ofk_launch_pyr3(", 1)
    open(os.path.join(tmp, "ofk_internal.h"), "w").write(hdr)
    api = open(os.path.join(tmp, "ofk_api.hip")).read()
    a = api.index("        if (!have_gray) {\n            StageTimer t(c, OFK_STAGE_GRAY, sa);\n            ofk_launch_gray(sa, bgr0")
    b = api.index("        if (overlap) {\n            OFK_HIP(c, hipEventRecord(c->ev_aux[k], sa));")
    new = '''        bool fusedp = false;
        if (!have_gray) {
            StageTimer t(c, OFK_STAGE_PYR, sa);
            fusedp = ofk_launch_gray_pyr3(sa, bgr0, nullptr, c->bgr_stride, pyr0, nullptr, c->pyr_stride, lv, nb, nb);
        }
        if (!have_gray && !fusedp) { StageTimer t(c, OFK_STAGE_GRAY, sa); ofk_launch_gray(sa, bgr0, c->bgr_stride, pyr0, c->pyr_stride, nb, h, w); }
        if (overlap) OFK_HIP(c, hipEventRecord(c->ev_g0[k], sa));
        if (fusedp) { StageTimer t(c, OFK_STAGE_PYR, sa); ofk_launch_gray_pyr3(sa, bgr1, nullptr, c->bgr_stride, pyr1, nullptr, c->pyr_stride, lv, nb, nb); }
        if (!have_gray && !fusedp) { StageTimer t(c, OFK_STAGE_GRAY, sa); ofk_launch_gray(sa, bgr1, c->bgr_stride, pyr1, c->pyr_stride, nb, h, w); }
        if (!fusedp) {
            StageTimer t(c, OFK_STAGE_PYR, sa);
            for (int l = ofk_launch_pyr3(sa, pyr0, pyr1, c->pyr_stride, lv, nb, 2 * nb) ? 4 : 1; l <= lv.n; ++l)
                ofk_launch_pyr_down2(sa, pyr0 + lv.off[l - 1], pyr1 + lv.off[l - 1], c->pyr_stride, lv.h[l - 1], lv.w[l - 1], pyr0 + lv.off[l],
                                     pyr1 + lv.off[l], c->pyr_stride, nb);
        } else {
            for (int l = 4; l <= lv.n; ++l)
                ofk_launch_pyr_down2(sa, pyr0 + lv.off[l - 1], pyr1 + lv.off[l - 1], c->pyr_stride, lv.h[l - 1], lv.w[l - 1], pyr0 + lv.off[l],
                                     pyr1 + lv.off[l], c->pyr_stride, nb);
        }
'''
    api = api[:a] + new + api[b:]
    open(os.path.join(tmp, "ofk_api.hip"), "w").write(api)
    subprocess.check_call(["make", "-s", "-C", CS])
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for f in sorted(os.listdir(CS)):
        if not f.endswith(".hip"): continue
        o = os.path.join(CS, "build", f[:-4] + ".o")
        if f in ("k_image.hip", "ofk_api.hip"):
            o = os.path.join(tmp, f[:-4] + ".o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, f"-I{tmp}", "-c", os.path.join(tmp, f), "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libofk_{name}.so"), *objs, "-ldl"])
    print("built", name)

if __name__ == "__main__":
    for f in os.listdir(OUT) if os.path.isdir(OUT) else []:
        if f.startswith("libofk_") and f.endswith(".so"): os.unlink(os.path.join(OUT, f))
    build("graypyr3w", True)
    build("graypyr3c", False)
    shutil.copy(os.path.join(ROOT, "drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd", "libofk.so"), os.path.join(OUT, "libofk_product.so"))

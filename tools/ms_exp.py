"""M-sweep kernel experiments: builds variant libraries (extra -D flags) and times the K = 256 -> M = 1024 shapes through each.
Usage (GPU box): python tools/ms_exp.py "name:-DFLAG -DFLAG2" "name2:..."   (name 'base' = the shipped library)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "da-sac_amd")
srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
objdir = os.path.join(PKG, "build")
modes = os.environ.get("MS_MODES", "fwd fwd_res_bits").split()
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    env = dict(os.environ)
    if flags.strip():
        out = "/tmp/libdasac_%s.so" % name
        # only the M-sweep source is rebuilt; the other objects come from the in-tree build
        o = "/tmp/ms_%s.o" % name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
                              + flags.split() + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-c",
                                                 os.path.join(PKG, "csrc", "gemm1x1_msweep.hip"), "-o", o], stderr=subprocess.DEVNULL)
        objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs if not s.endswith("gemm1x1_msweep.hip")] + [o]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
        env["DASAC_LIB"] = out
    for m in modes:
        shape = "l3_1x1a" if m.startswith("dgrad") else "l3_1x1b"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "one_conv.py"), shape, m, "20", "16"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("done")]
        print("{:18s} {}".format(name, line[0][5:] if line else "FAILED " + r.stderr[-300:]), flush=True)

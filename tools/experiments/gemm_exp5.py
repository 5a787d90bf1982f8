"""conv_gemm experiments of round 5: builds variant libraries (extra -D flags on conv_igemm.hip) and times single shapes through each.
Usage (GPU box): python tools/gemm_exp5.py "name:-DFLAG ..." ...      ('base:' = the shipped library); DASAC_MSWEEP=0 is forced."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "da-sac_amd")
srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
objdir = os.path.join(PKG, "build")
shapes = [s.split(":") for s in os.environ.get("EXP_SHAPES", "l3_1x1b:fwd l3_1x1b:fwd_res l3_1x1a:fwd l3_3x3:fwd").split()]
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    env = dict(os.environ, DASAC_MSWEEP="0")
    src = os.path.join(PKG, "csrc", "conv_igemm.hip")
    if flags.startswith("@"):          # "@path" = another version of the source file (experiments against an older loop)
        path, _, flags = flags[1:].partition(" ")
        src = os.path.join(ROOT, path)
    if flags.strip() or src != os.path.join(PKG, "csrc", "conv_igemm.hip"):
        out, o = "/tmp/libdasac_%s.so" % name, "/tmp/ci_%s.o" % name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
                              + flags.split() + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-c", src, "-o", o], stderr=subprocess.DEVNULL)
        objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs if not s.endswith("conv_igemm.hip")] + [o]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
        env["DASAC_LIB"] = out
    for sh in shapes:
        shape, m, B = sh[0], sh[1], (sh[2] if len(sh) > 2 else "16")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "one_conv.py"), shape, m, "20", B], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("done")]
        print("{:14s} {}".format(name, line[0][5:].split("checksum")[0] if line else "FAILED " + r.stderr[-300:]), flush=True)

"""Head-kernel experiments: builds variant libraries (extra -D flags on head.hip) and times one tools/head_bw.py row through each.
Usage (GPU box): HEAD_BW_ONLY="low-res" python tools/head_exp.py "name:-DFLAG ..." ...      ('base:' = the shipped library)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "da-sac_amd")
srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
objdir = os.path.join(PKG, "build")
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    env = dict(os.environ)
    src = os.path.join(PKG, "csrc", "head.hip")
    if flags.startswith("@"):          # "@path" = another version of head.hip
        path, _, flags = flags[1:].partition(" ")
        src = os.path.join(ROOT, path)
    if flags.strip() or src != os.path.join(PKG, "csrc", "head.hip"):
        out, o = "/tmp/libdasac_%s.so" % name, "/tmp/head_%s.o" % name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
                              + flags.split() + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-c", src, "-o", o], stderr=subprocess.DEVNULL)
        objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs if not s.endswith("head.hip")] + [o]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
        env["DASAC_LIB"] = out
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "head_bw.py")], env=env, capture_output=True, text=True)
    for l in r.stdout.splitlines():
        if l.rstrip().endswith("%"):
            print("{:14s} {}".format(name, l), flush=True)
    if r.returncode:
        print(name, "FAILED", r.stderr[-400:], flush=True)

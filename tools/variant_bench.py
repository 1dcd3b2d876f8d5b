import sys, os, subprocess, json, glob
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [None] + sorted(glob.glob(os.path.join(root, "badger_amcl_amd", "libexp_*.so")))
for lib in libs:
    env = dict(os.environ)
    if lib: env["BPF_LIB"] = lib
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "5", "--cpu-budget", "0"],
                         env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(os.path.basename(lib) if lib else "default", "step ms %.4f" % d["ms_per_step"], "score us %.1f" % (d["roofline"]["kernel_ms"]*1e3), d["kernel_ms_per_step"])
    except Exception as ex:
        print(lib, "FAILED", out.stderr[-500:])

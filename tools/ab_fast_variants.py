import os, sys, subprocess, re, collections
L = os.path.join(os.getcwd(), "imagemagick_amd/lib")
variants = sys.argv[1:]
res = collections.defaultdict(list)
for rnd in range(3):
    for v in variants:
        env = dict(os.environ, TBQ_REPS="60")
        if v != "new":
            env["MAGICKHIP_LIBRARY"] = "%s/libmagickhip_%s.so" % (L, v)
        else:
            env.pop("MAGICKHIP_LIBRARY", None)
        out = subprocess.run([sys.executable, "tools/time_blur_quick.py", "8192", "10"], env=env, capture_output=True, text=True).stdout
        for l in out.splitlines():
            m = re.search(r"^(\w+)\s+sigma 10\s+fast .*blur_fused_hybrid': '([0-9.]+) \[([0-9.]+)\]", l)
            if m:
                res[(m.group(1), v)].append((float(m.group(2)), float(m.group(3))))
for k in sorted(res):
    print("%-7s %-5s avg %s  min %s" % (k[0], k[1], " ".join("%.4f" % a for a, _ in res[k]), " ".join("%.4f" % b for _, b in res[k])))

import os, struct, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
ge.build()
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exe = os.path.join(ROOT, "mcmc-ammsb-gpu_amd", "ammsb_main")
d = tempfile.mkdtemp()
f = os.path.join(d, "g.bin.gz")
hostlib.dump_dataset(f, 270000, 0.01, hostlib.generate_graph(270000, 16, 12, seed=3))
common = [exe, "--load-data", "1", "--load-file", f, "-k", "1024", "-m", "2048", "-n", "16", "-x", "4", "-i", "2",
          "--phi-wg", "64", "--beta-wg", "64", "--ppx-wg", "64", "--device-sampling", "1", "--async", "1", "--graph", "1"]
def records(data):
    recs, pos = [], 0
    while pos < len(data):
        (n,) = struct.unpack_from("<Q", data, pos)
        recs.append(data[pos + 8:pos + 8 + n]); pos += 8 + n
    return recs
outs = []
for cands in ("0", "3", "0"):
    ck = os.path.join(d, "c.ckpt")
    r = subprocess.run(common + ["--pi-candidates", cands, "--checkpoint-out", ck], capture_output=True, text=True)
    print(cands, r.returncode, [l for l in r.stderr.splitlines() if "placement" in l or "ppx" in l][:6])
    outs.append(records(open(ck, "rb").read()))
for j in (1, 2):
    print("run 0 vs run", j)
    for i, (a, b) in enumerate(zip(outs[0], outs[j])):
        if a != b:
            k = next(x for x in range(min(len(a), len(b))) if a[x] != b[x])
            nd = sum(1 for x in range(min(len(a), len(b))) if a[x] != b[x]) if len(a) < 300000 else -1
            print("  record", i, "len", len(a), len(b), "first diff at", k, "n diff", nd, a[k:k+16].hex(), b[k:k+16].hex())
print("sizes", [len(a) for a in outs[0]])

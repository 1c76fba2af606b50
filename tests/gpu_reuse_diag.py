import sys, os
sys.path.insert(0, "tests")
import numpy as np
import gpu_common as gc
from oracle_lib import parse_stream
n0, dim = 1200, 16
mb, gd = 64, 4
X, Q = gc.make_data(2400, dim, "l2sq", 777)
cpu, gpu = gc.oracle_index(dim, "l2sq", 8, 16, 40), gc.gpu_index(dim, "l2sq", 8, 16, 40)
cpu.reserve(4096), gpu.reserve(4096)
gpu.set_build_params(mb, gd)
cpu.build_batch(np.arange(n0), X[:n0], mb, gd); gpu.add(np.arange(n0), X[:n0])
entry_key = cpu.node_key(cpu.entry_slot())
dead = sorted(set([entry_key] + list(range(5, 250, 5))))
gpu.remove(np.asarray(dead, dtype=np.int64))
for k in dead: cpu.remove(int(k))
nadd = int(sys.argv[1]) if len(sys.argv) > 1 else 80
keys = 10_000 + np.arange(nadd)
cpu.build_batch(keys, X[n0:n0 + nadd], mb, gd); gpu.add(keys, X[n0:n0 + nadd])
a, b = parse_stream(gpu.save()), parse_stream(cpu.save())
print("keys equal", np.array_equal(a["keys"], b["keys"]), "vectors equal", np.array_equal(a["vectors"], b["vectors"]))
reused = set(np.nonzero(a["keys"][:n0] >= 10_000)[0].tolist())
print("reused slots", sorted(reused)[:60])
for s in range(a["rows"]):
    for l in range(len(a["adj"][s])):
        if not np.array_equal(a["adj"][s][l], b["adj"][s][l]):
            print("slot", s, "reused" if s in reused else ("new" if s >= n0 else "old"), "level", l, "gpu", a["adj"][s][l].tolist(), "cpu", b["adj"][s][l].tolist())

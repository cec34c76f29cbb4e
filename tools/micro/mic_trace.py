"""Where a hand-off of the forward MIC sweep spends its time: python tools/micro/mic_trace.py (needs tools/micro/_abl/libmanta_trace.so, the library
built with -DROWS_TRACE=1).  Stamps are 100 MHz wall-clock ticks: 0 = producer's compute wave has a half block in the ring, 1 = its publisher
has issued the granules, 2 = consumer's poller has seen them, 3 = consumer's compute wave starts the block."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mantaflow_amd import _lib, core

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
path = os.path.join(ROOT, "tools", "micro", "_abl", sys.argv[2] if len(sys.argv) > 2 else "libmanta_trace.so")
_lib.use_library(path, "cuda")
s = core.Solver(name="m", gridSize=core.vec3(n, n, n))
lib = s.lib
flags = core.FlagGrid(s)
flags.initDomain(boundaryWidth=0)
flags.fillGrid()
A0, Ai, Aj, Ak, ap, dst, src = (core.Grid(s) for _ in range(7))
lib.call("mf_make_laplace_matrix", n, n, n, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, s.stream)
src.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (n, n, n)).astype(np.float32))
lib.call("mf_mic_init", n, n, n, flags.ptr, ap.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
for _ in range(4):
    lib.call("mf_mic_apply", n, n, n, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
torch.cuda.synchronize()
raw = ctypes.CDLL(path)
NB, NBLK = 1024, 40
buf = np.zeros(NB * NBLK * 4, np.int64)
rc = raw.mf_debug_mic_trace(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
T = buf.reshape(NB, NBLK, 4).astype(np.float64) * 10.0      # ns
nb = n // 8
nblk = n // 8 + 2
t0 = T[0, 0, 3]
print("sweep: first block starts at 0, last bundle's last block starts at %.1f us" % ((T[nb * nb - 1, nblk - 1, 3] - t0) / 1e3))
pace = []
for sid in range(nb * nb):
    e0 = T[sid, 1:nblk - 1, 0]
    pace.append(np.median(np.diff(e0)) / 8.0)
print("pace of a bundle: median %.1f ns per step (10 %%: %.1f, 90 %%: %.1f)" % (np.median(pace), np.percentile(pace, 10), np.percentile(pace, 90)))
d_pub, d_mem, d_cons, d_tot, lag = [], [], [], [], []
for tk in range(nb):
    for tj in range(nb):
        sid = tk * nb + tj
        prods = ([sid - 1] if tj > 0 else []) + ([sid - nb] if tk > 0 else [])
        if not prods:
            continue
        for m in range(2, nblk - 3):
            e0 = max(T[p, m + 1, 0] for p in prods)
            plast = max(prods, key=lambda p: T[p, m + 1, 1])
            e1 = T[plast, m + 1, 1]
            e2, e3 = T[sid, m, 2], T[sid, m, 3]
            if e3 - e2 > 400:          # the consumer was busy with its previous block: not a hand-off on the critical path
                continue
            d_pub.append(T[plast, m + 1, 1] - T[plast, m + 1, 0])
            d_mem.append(e2 - e1)
            d_cons.append(e3 - e2)
            d_tot.append(e3 - e0)
            lag.append(e3 - max(T[p, m, 3] for p in prods))
for name, v in (("producer: half block in the ring -> granules issued", d_pub), ("granules issued -> consumer's poller has them", d_mem),
                ("poller -> consumer's compute wave starts the block", d_cons), ("total: producer half block -> consumer block start", d_tot),
                ("consumer block start behind producer's start of the same block", lag)):
    v = np.asarray(v)
    print("%-70s median %6.0f ns   10 %% %6.0f   90 %% %6.0f   (%d samples)" % (name, np.median(v), np.percentile(v, 10), np.percentile(v, 90), len(v)))

#!/usr/bin/env python3
"""Event model of one MIC 'rows' sweep: bundles (tj,tk) on an nb x nb grid, bundle streaming time T, hand-off lag to a consumer
hop_local (same XCD) or hop_remote; WG pools per queue.  Used to choose the diagonal-band partition of bundles over XCDs."""
import heapq, sys, itertools

def simulate(nb, bounds, wg_per_q, T=48.0, hl=3.5, hr=4.4, single=False):
    # queue of a bundle: band of d = tj - tk by bounds (sorted list of upper bounds, len nq-1)
    def q_of(tj, tk):
        if single: return 0
        d = tj - tk
        for i, b in enumerate(bounds):
            if d < b: return i
        return len(bounds)
    nq = 1 if single else len(bounds) + 1
    queues = [[] for _ in range(nq)]
    for L in range(2 * nb - 1):
        for tk in range(max(0, L - nb + 1), min(nb, L + 1)):
            tj = L - tk
            queues[q_of(tj, tk)].append((tj, tk))
    start, end = {}, {}
    # each queue: WG free times heap; bundles drawn in order by the earliest-free WG
    free = [[0.0] * (wg_per_q if not single else wg_per_q * 8) for _ in range(nq)]
    for f in free: heapq.heapify(f)
    ptr = [0] * nq
    done = 0
    total = nb * nb
    # process in global L order is not valid across queues (draw times differ); iterate: pick the queue whose next draw is earliest
    # but start time depends on preds that may not be computed yet -> loop until all resolved
    pending = True
    while done < total:
        progressed = False
        for q in range(nq):
            while ptr[q] < len(queues[q]):
                tj, tk = queues[q][ptr[q]]
                preds = [p for p in ((tj - 1, tk), (tj, tk - 1)) if p[0] >= 0 and p[1] >= 0]
                if any(p not in start for p in preds): break
                t0 = heapq.heappop(free[q])
                s = t0
                e_min = 0.0
                for p in preds:
                    hop = hr if (single or q_of(*p) != q) else hl
                    s = max(s, start[p] + hop)
                    e_min = max(e_min, end[p] + hop)
                e = max(s + T, e_min)
                start[(tj, tk)], end[(tj, tk)] = s, e
                heapq.heappush(free[q], e)
                ptr[q] += 1
                done += 1
                progressed = True
        assert progressed
    return max(end.values()), [len(x) for x in queues]

if __name__ == "__main__":
    nb = 32
    print("single queue 256 WGs:", simulate(nb, [], 32, single=True)[0])
    best = None
    # symmetric bounds: d thresholds  -a3,-a2,-a1,0,a1,a2,a3  (+1 shift for upper-bound semantics)
    for a1, a2, a3 in itertools.combinations(range(1, 24), 3):
        bounds = [-a3 + 1, -a2 + 1, -a1 + 1, 1, a1 + 1, a2 + 1, a3 + 1]
        bounds = [-a3, -a2, -a1, 0, a1, a2, a3]
        m, sizes = simulate(nb, bounds, 32)
        if best is None or m < best[0]:
            best = (m, bounds, sizes)
    print("best symmetric bands:", best)
    for hl in (3.0, 3.5, 4.0):
        print(hl, simulate(nb, best[1], 32, hl=hl)[0])

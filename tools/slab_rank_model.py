"""What ONE rank of an N-rank z-slab run computes per 256^3 smoke step, measured on one GPU: the slab layer runs with a stand-in
communicator of `world` ranks whose exchanges do nothing (ghost planes keep what they hold) and whose gathers return the rank's own row
`world` times (alpha = sigma / dp and beta keep their ratios, so the local PCG converges like a stand-alone solve of the slab).  Everything
but the communication is the code of a real rank: advection on owned + ghost planes, the PCG on the 1-ghost window with the y / x blocked
slab-local MIC(0), the one-launch alpha / beta kernels.  Gives the device time per step and per PCG iteration that the collectives of a
real run come on top of.   python tools/slab_rank_model.py [world=8] [rank=3] [steps=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from mantaflow_amd import core, slab


class StandInComm(object):
    def __init__(self, world, rank):
        self.on, self.world, self.rank, self.stage = True, world, rank, False

    def sendrecv(self, pairs):
        pass

    def gather_scalars(self, vals, device):
        return np.tile(np.asarray([vals], np.float64), (self.world, 1))

    def allgather_dev(self, t):
        return t.unsqueeze(0).expand(self.world, *t.shape).contiguous()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else min(3, world - 1)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    n, dt = 256, 1.0
    torch.cuda.set_device(0)
    G = slab.required_ghost(2.0)
    dom = slab.SlabDomain((n, n, n), G, comm=StandInComm(world, rank))
    s = dom.solver
    s.timestep = dt
    flags, vel, vel0, dens, pres = core.FlagGrid(s), core.MACGrid(s), core.MACGrid(s), core.Grid(s), core.Grid(s)
    dom.scatter_global(flags, slab.global_flags(n))
    dom.scatter_global(vel0, bench.synthetic_velocity(n, n, n))
    slab.setWallBcs(dom, flags, vel0)
    dom.scatter_global(dens, bench.synthetic_density(n, n, n))
    stats, its, ms, ms_solve = {}, [], [], []
    for it in range(steps + 1):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        # slab.smoke_step, with an event in front of the solve
        vel.copyFrom(vel0)
        dom.exchange(dens)
        slab.advectSemiLagrange(dom, flags, vel, dens, order=2)
        slab.advectSemiLagrange(dom, flags, vel, vel, order=2)
        dom.exchange(vel, 1)
        slab.setWallBcs(dom, flags, vel)
        e1.record()
        slab.solvePressure(dom, vel, pres, flags, stats=stats)
        e2.record()
        torch.cuda.synchronize()
        if it > 0:
            ms.append(e0.elapsed_time(e2))
            ms_solve.append(e1.elapsed_time(e2))
            its.append(stats["iterations"])
    # the PCG alone: time a second solve of the same system shape through solvePressure
    print("rank %d of %d: planes owned %d (+%d / +%d ghosts), PCG window %d planes, MIC blocks %s" %
          (rank, world, dom.nown, dom.gl, dom.gu, dom.nown + (1 if dom.gl else 0) + (1 if dom.gu else 0), stats.get("mic_blocking")))
    print("device time per step %.2f ms (advection + boundary conditions %.2f ms), solvePressure %.2f ms for %s iterations of the stand-alone slab "
          "-> %.0f us per PCG iteration incl. the set-up" %
          (float(np.mean(ms)), float(np.mean(ms)) - float(np.mean(ms_solve)), float(np.mean(ms_solve)), its,
           1e3 * float(np.mean(ms_solve)) / max(1.0, float(np.mean(its)))))


if __name__ == "__main__":
    main()

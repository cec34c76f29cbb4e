"""z-slab decomposition of the smoke hot path across the GPUs of one node (one process per GPU, torch.distributed).

The reference has no multi-device code at all, so this layer is new design:

* rank r owns the z-planes [z0, z1) of the global grid and keeps G ghost planes on each interior side; its *local*
  grid (owned + ghosts, clipped at the domain walls) is an ordinary single-device grid, so every kernel of the
  C ABI runs on it unchanged.  Results are valid on the owned planes (and on as many ghost planes as the kernel's
  reach leaves intact); ghosts are refreshed by `exchange` = point-to-point send/recv with the two z-neighbours only
  (2 of the 7 xGMI links; a 256^2 fp32 plane is 256 KiB).
* advection: G = 2R planes with R = ceil(max|v_z| dt) + 1 (the reach of one semi-Lagrangian gather); with the ghosts
  filled once, MacCormack (forward trace, backward trace from the forward field, correction, clamp) is entirely
  local and bit-identical to the single-device result on the owned planes.
* pressure: PCG with one-plane halo exchange of the search vector per iteration and the CG scalars combined with ONE
  all-gather per reduction point (8-byte payloads: latency, not bandwidth, so no ring all-reduce).  The reference's
  MIC(0) sweep has a k-1 dependency that spans slabs; here each slab applies MIC(0) of its own diagonal block
  (block-Jacobi: the coupling Ak across slab faces is dropped in the preconditioner only).  With one rank this is
  exactly the reference algorithm; with P > 1 the iterates differ and the result is validated at converged-solution
  level (same stopping rule max|residual| < cgAccuracy on the true residual).
No data-path collective other than these; FLIP particles across slabs (reverse halo + migration) are "next".
"""
import ctypes
import os
import math

import numpy as np
import torch
import torch.distributed as dist

from . import core
from .core import _ptr


STOP_POLL = 4          # slab PCG: iterations between two host reads of the device-side stop state
MIC_BLOCK_CELLS_X = int(os.environ.get("MF_SLAB_XBLOCK", "128"))   # x-extent of a preconditioner block for P > 1 (0 = whole rows)
MIC_BLOCK_ROWS = int(os.environ.get("MF_SLAB_YBLOCK", "64"))   # y-extent of a preconditioner block when the domain is split over several ranks (0 = do not cut)


def _off(t, nelem):
    return ctypes.c_void_p(t.data_ptr() + 4 * int(nelem))


class Comm(object):
    """thin wrapper: p2p plane exchange + small all-gathers; stages through the host when the backend cannot move
    device tensors (gloo rehearsals of the GPU path); a world of 1 needs no process group."""

    def __init__(self):
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.stage = self.on and dist.get_backend() == "gloo"

    def sendrecv(self, pairs):
        """pairs: list of (send_tensor|None, recv_tensor|None, peer) -- all posted together"""
        if not self.on:
            return
        ops, post = [], []
        for snd, rcv, peer in pairs:
            if snd is not None:
                b = snd.contiguous()
                if self.stage and b.is_cuda:
                    b = b.cpu()
                ops.append(dist.P2POp(dist.isend, b, peer))
            if rcv is not None:
                if self.stage and rcv.is_cuda:
                    tmp = torch.empty(rcv.shape, dtype=rcv.dtype)
                    ops.append(dist.P2POp(dist.irecv, tmp, peer))
                    post.append((rcv, tmp))
                elif rcv.is_contiguous():
                    ops.append(dist.P2POp(dist.irecv, rcv, peer))
                else:
                    tmp = torch.empty(rcv.shape, dtype=rcv.dtype, device=rcv.device)
                    ops.append(dist.P2POp(dist.irecv, tmp, peer))
                    post.append((rcv, tmp))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        for rcv, tmp in post:
            rcv.copy_(tmp)

    def gather_scalars(self, vals, device):
        """all ranks get [world][len(vals)] float64 (summed / maxed by the caller in rank order -> identical everywhere)"""
        if not self.on:
            return np.asarray([vals], np.float64)
        t = torch.tensor(vals, dtype=torch.float64, device="cpu" if self.stage else device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return torch.stack(out).cpu().numpy()


    def allgather_dev(self, t):
        """[k] tensor on the compute device -> [world][k] on the same device, without touching the host when the backend
        moves device memory itself (nccl = RCCL: ONE collective into one tensor, no per-rank list and no stacking copy); rows are
        in rank order on every rank, so reductions over them are bit-identical everywhere.  A world of one gathers nothing."""
        if not self.on or self.world == 1:
            return t.unsqueeze(0)
        if dist.get_backend() != "nccl":
            # gloo rehearsals (CPU tensors, or device tensors staged through the host)
            h = t.cpu() if t.is_cuda else t
            out = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(out, h)
            return torch.stack(out).to(t.device)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out


class SlabDomain(object):
    def __init__(self, gsize, ghost, comm=None, zrange=None):
        """zrange = (z0, z1): this rank's owned planes, when they are not the even split (the fine solver of a two-resolution
        scene owns upres x the coarse solver's planes: `refine`)"""
        self.comm = comm or Comm()
        self.NX, self.NY, self.NZ = (int(v) for v in gsize)
        P, r = self.comm.world, self.comm.rank
        if zrange is None:
            self.z0, self.z1 = slab_ranges(self.NZ, P)[r]
        else:
            self.z0, self.z1 = int(zrange[0]), int(zrange[1])
        self.G = int(ghost)
        if P > 1 and (self.z1 - self.z0) < self.G:
            raise RuntimeError("slab of %d planes is thinner than the ghost width %d" % (self.z1 - self.z0, self.G))
        self.lo = max(0, self.z0 - self.G)
        self.hi = min(self.NZ, self.z1 + self.G)
        self.gl = self.z0 - self.lo         # ghost planes below / above
        self.gu = self.hi - self.z1
        self.nown = self.z1 - self.z0
        self.LZ = self.hi - self.lo
        self.XY = self.NX * self.NY
        self.solver = core.Solver(gridSize=core.vec3(self.NX, self.NY, self.LZ), dim=3)
        self.solver._global_size = (self.NX, self.NY, self.NZ)         # getDx() / noise scaling / resampling factors of the undivided domain
        # positions handed to the interpolators are global coordinates -> bit-identical to the undivided domain.  The window
        # belongs to THIS solver (core.SolverLib sets it per call): other solvers of the process keep their own
        self.solver._slab_window = (self.lo, self.NZ)
        self.below = r - 1 if r > 0 else None
        self.above = r + 1 if r < P - 1 else None

    # -- views ------------------------------------------------------------------------------------------
    def planes(self, grid, k0, k1):
        """[ncomp][k1-k0][NY*NX] view of local planes k0..k1 of a grid"""
        return grid.data.view(grid._ncomp, self.LZ, self.XY)[:, k0:k1]

    def own_off(self):
        return self.gl * self.XY

    @property
    def n_own(self):
        return self.nown * self.XY

    def scatter_global(self, grid, arr):
        """fill the local grid (owned + ghosts) from a global SoA numpy array [ncomp][NZ][NY][NX] / [NZ][NY][NX]"""
        a = arr[:, self.lo:self.hi] if grid._ncomp == 3 else arr[self.lo:self.hi]
        grid.data.copy_(torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(grid.data.device))

    def gather_owned(self, grid):
        a = self.planes(grid, self.gl, self.gl + self.nown).detach().cpu().numpy().copy()
        a = a.reshape(grid._ncomp, self.nown, self.NY, self.NX)
        return a if grid._ncomp == 3 else a[0]

    # -- ghost exchange -----------------------------------------------------------------------------------
    def exchange(self, grid, width=None):
        """fill my ghost planes (up to `width` on each side) from the neighbours' owned planes"""
        w = self.G if width is None else int(width)
        pairs = []
        if self.below is not None:
            wl = min(w, self.gl)
            pairs.append((self.planes(grid, self.gl, self.gl + w), self.planes(grid, self.gl - wl, self.gl), self.below))
        if self.above is not None:
            wu = min(w, self.gu)
            top = self.gl + self.nown
            pairs.append((self.planes(grid, top - w, top), self.planes(grid, top, top + wu), self.above))
        self.comm.sendrecv(pairs)

    # -- global reductions over owned cells ---------------------------------------------------------------
    def max_abs_owned(self, grid, comp=None):
        s = self.solver
        r = ctypes.c_float()
        base = 0 if comp is None else comp * grid.n
        s.lib.call("mf_grid_max_abs", self.n_own, _off(grid.data, base + self.own_off()), ctypes.byref(r), s.stream)
        g = self.comm.gather_scalars([r.value], s.device)
        return float(np.max(g[:, 0]))


def refine(dom, upres, ghost=None):
    """the slab of an `upres` times finer solver on the same z-range (waveletTurbulence.py's `xl` solver next to `sm`): fine
    plane K lies in coarse plane K // upres, so every rank owns the fine planes of its coarse planes"""
    u = int(upres)
    return SlabDomain((dom.NX * u, dom.NY * u, dom.NZ * u), dom.G if ghost is None else ghost, comm=dom.comm, zrange=(dom.z0 * u, dom.z1 * u))


def required_ghost(maxvz_dt):
    """G = 2R, R = ceil(max|v_z| dt) + 1 (reach of one trilinear gather after a trace of that length)"""
    return 2 * (int(math.ceil(maxvz_dt)) + 1)


# =========================================================================================================
# distributed plugins (names mirror the single-device ones)
# =========================================================================================================
def has_outflow(dom, flags):
    """does the DOMAIN hold an outflow cell (setOpenBound)?  Looked at once per flag grid: outflow cells are scene set-up."""
    key = (flags.data.data_ptr(),)
    if getattr(flags, "_slab_outflow_key", None) != key:
        dom.solver.sync()
        mine = float(bool((flags.data & core.TypeOutflow).any().item()))
        flags._slab_outflow = bool(np.max(dom.comm.gather_scalars([mine], dom.solver.device)) > 0)
        flags._slab_outflow_key = key
    return flags._slab_outflow


def advectSemiLagrange(dom, flags, vel, grid, order=1, strength=1.0, clampMode=2):
    """advection.cpp:293-322 / 407-437 on a slab; ghosts of `vel` and `grid` must be current (dom.exchange).  MAC grids with
    outflow cells in the domain: applyOutflowBC (advection.cpp:347-392) reads the ADVECTED velocity up to two cells around an
    outflow cell, so it runs after the advected field's ghosts have been fetched (two planes), not inside the local advection."""
    from . import plugins
    s = dom.solver
    m = dom.max_abs_owned(vel, comp=2) * s.getDt()
    if dom.comm.world > 1 and required_ghost(m) > dom.G:
        raise RuntimeError("slab advection: max|v_z| dt = %.2f needs %d ghost planes, domain has %d" % (m, required_ghost(m), dom.G))
    split_bc = dom.comm.world > 1 and bool(grid.getType() & core.GridBase.TypeMAC) and has_outflow(dom, flags)
    if not split_bc:
        plugins.advectSemiLagrange(flags, vel, grid, order=order, strength=strength, clampMode=clampMode)
        return
    prev = core.MACGrid(s)
    prev.copyFrom(grid)                         # velPrev of the convective BC: the field before this advection
    flags._may_have_outflow = False
    try:
        plugins.advectSemiLagrange(flags, vel, grid, order=order, strength=strength, clampMode=clampMode)
    finally:
        del flags._may_have_outflow
    dom.exchange(grid, 2)
    plugins._apply_outflow_bc(flags, grid, prev, s.getDt())


def solvePressure(dom, vel, pressure, flags, cgAccuracy=1e-3, cgMaxIterFac=1.5, stats=None, phi=None, gfClamp=1e-04):
    """computePressureRhs + PCG (slab-local MIC) + correctVelocity on a slab.  vel ghosts (1 plane) must be current; with `phi`
    (ghost-fluid free surface, pressure.cpp:136-214) its one-plane ghosts as well."""
    s = dom.solver
    lib, st = s.lib, s.stream
    sx, sy, sz = dom.NX, dom.NY, dom.LZ
    n, XY, off, nown = sx * sy * sz, dom.XY, dom.own_off(), dom.n_own
    G = core.Grid
    rhs, residual, search, tmp, A0, Ai, Aj, Ak, Akm, Ap = (G(s) for _ in range(10))
    lib.call("mf_make_rhs", sx, sy, sz, flags.ptr, rhs.ptr, vel.ptr, None, None, None, None, None, 0.0, 1e-4, None, None, st)
    lib.call("mf_make_laplace_matrix", sx, sy, sz, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, st)
    if phi is not None:
        # ApplyGhostFluidDiagonal (pressure.cpp:136-151): the diagonal of a fluid cell next to an empty one, from phi of the two
        lib.call("mf_apply_ghost_fluid_diagonal", sx, sy, sz, A0.ptr, flags.ptr, phi.ptr, float(gfClamp), st)
    # The PCG runs on a WINDOW of the local grid: the owned planes plus ONE ghost plane per interior face (ApplyMatrix reads search at
    # +-1 plane; everything else is local).  The advection ghosts beyond (G = 6 at max|v| dt = 2: 44 planes for 32 owned ones at
    # 8 ranks) would only be swept and streamed for nothing.  Slabs are z-contiguous, so the window is a pointer offset.
    w0, w1 = max(dom.gl - 1, 0), min(dom.gl + dom.nown + 1, sz)
    wz, wgl = w1 - w0, dom.gl - w0                      # planes of the window, ghost planes below the owned ones inside it (0 or 1)
    woff = w0 * XY                                       # first cell of the window in the local grids
    off_w = wgl * XY                                     # first owned cell inside the window

    def W(g):
        return _off(g.data, woff)

    # slab-local MIC: ghost planes are not part of the block, the coupling across the slab faces is cut
    fmic = core.FlagGrid(s)
    fmic.copyFrom(flags)
    fv = fmic.data.view(sz, XY)
    if dom.gl:
        fv[:dom.gl] = core.TypeObstacle
    if dom.gu:
        fv[dom.gl + dom.nown:] = core.TypeObstacle
    # ApplyMatrix runs with these flags too: on the ghost planes it then passes `search` through instead of applying the stencil
    # (nothing there is needed; the outermost ghost plane has no neighbour plane to read, and the fused dot products below
    # multiply those values with the zero residual of the ghost planes), the owned planes are unaffected (the stencil never
    # looks at a neighbour's flags)
    lib.call("mf_pack_matrix", sx, sy, wz, W(fmic), W(A0), W(Ai), W(Aj), W(Ak), st)       # 9 (13 with a ghost-fluid diagonal) instead of 28 B per cell
    Akm.copyFrom(Ak)
    av = Akm.data.view(sz, XY)
    if dom.gl:
        av[dom.gl - 1] = 0
    if dom.gu:
        av[dom.gl + dom.nown - 1] = 0
    # with more than one rank the preconditioner is block-Jacobi anyway: also cut it into blocks of MIC_BLOCK_ROWS rows
    # along y, so that the sweeps of a (thin) slab are not one long dependency chain through all of y
    Ajm = Aj
    jblock = 0
    if dom.comm.world > 1 and MIC_BLOCK_ROWS > 0 and sy >= 2 * MIC_BLOCK_ROWS:
        jblock = MIC_BLOCK_ROWS
        Ajm = G(s)
        Ajm.copyFrom(Aj)
        ajv = Ajm.data.view(sz, sy, sx)
        for jc in range(jblock, sy, jblock):
            ajv[:, jc - 1, :] = 0
    # ... and along x (blocks of MIC_BLOCK_CELLS_X cells): a bundle of rows then streams half a row, two to a row
    Aim = Ai
    xblock = 0
    if dom.comm.world > 1 and MIC_BLOCK_CELLS_X > 0 and sx >= 2 * MIC_BLOCK_CELLS_X:
        xblock = MIC_BLOCK_CELLS_X
        Aim = G(s)
        Aim.copyFrom(Ai)
        aiv = Aim.data.view(sz, sy, sx)
        for ic in range(xblock, sx, xblock):
            aiv[:, :, ic - 1] = 0
    # rhs must vanish outside the owned planes for the local sweeps to be well defined
    rv = rhs.data.view(sz, XY)
    if dom.gl:
        rv[:dom.gl] = 0
    if dom.gu:
        rv[dom.gl + dom.nown:] = 0
    gmax = max(dom.NX, dom.NY, dom.NZ)
    maxIter = int(np.float32(cgMaxIterFac) * np.float32(gmax))

    dev = s.device
    world = dom.comm.world
    red = torch.zeros(2, dtype=torch.float64, device=dev)     # {max|residual|, dot} of this rank
    # the scalar block of include/manta_hip.h (CgScalars layout): sigma, alpha, nalpha, beta, resNorm, ... xpending (fp32 like the
    # reference's Real members)
    sc = torch.zeros(16, dtype=torch.float32, device=dev)
    p_sc = ctypes.c_void_p(sc.data_ptr())
    p_sigma, p_alpha, p_beta, p_res = (ctypes.c_void_p(sc.data_ptr() + 4 * i) for i in (0, 1, 3, 4))
    p_red0, p_red1 = ctypes.c_void_p(red.data_ptr()), ctypes.c_void_p(red.data_ptr() + 8)

    def dot_own(a, b):
        lib.call("mf_grid_dot_dev", nown, _off(a.data, off), _off(b.data, off), p_red1, st)

    def mic(dst, src):
        lib.call("mf_mic_apply", sx, sy, wz, W(fmic), W(dst), W(src), W(Ap), W(Aim), W(Ajm), W(Akm), st)

    # All scalars stay on the device (fp32 like the reference's Real members); each reduction point is ONE all-gather whose
    # rows are combined in rank order by a one-thread kernel; the host reads one number per STOP_POLL iterations (the stopping test).
    # doInit, conjugategrad.cpp:210-235
    pressure.clear()
    residual.copyFrom(rhs)
    # the y / x blocking travels with this system (flags / Aprecond / Aj / Ak pointers), not with the process
    lib.call("mf_mic_init_blocked", sx, sy, wz, W(fmic), W(Ap), W(A0), W(Aim), W(Ajm), W(Akm), jblock, xblock, st)
    mic(tmp, residual)
    search.copyFrom(tmp)
    dot_own(tmp, residual)
    sc[0] = 1.0                                                  # sigma := (Real)sum via the beta step (beta unused here)
    g0 = dom.comm.allgather_dev(red)          # (kept referenced until the kernel that reads it has been queued behind it)
    lib.call("mf_cg_slab_beta", _ptr(g0), world, p_sigma, p_beta, p_res, 0.0, 0, None, st)
    # The stopping test runs on the device (state = {stop, iteration}); the host looks at it every STOP_POLL iterations.
    # Iterations queued past the stop are no-ops for pressure and residual (alpha = 0, no pending update), so the result is the
    # one the reference stops with, and the host never drains the stream inside the loop.
    state = torch.zeros(2, dtype=torch.int32, device=dev)
    p_state = ctypes.c_void_p(state.data_ptr())
    acc32 = float(np.float32(cgAccuracy))
    keep = []
    iters, resNorm, stop = 0, 1e20, 0
    # the host looks at the stop state one batch of STOP_POLL iterations BEHIND what it has queued (pinned copies + events), so the
    # stream never runs dry while Python queues the next batch; iterations queued past the stop are no-ops
    on_gpu = state.is_cuda
    hstate = [torch.zeros(2, dtype=torch.int32).pin_memory() if on_gpu else None for _ in range(2)]
    hev = [torch.cuda.Event() if on_gpu else None for _ in range(2)]
    slot, pending = 0, -1
    for it in range(1, maxIter + 1):
        dom.exchange(search, 1)
        # tmp = A search with dot(tmp, search) over the owned planes fused in
        lib.call("mf_apply_matrix_dot_dev", sx, sy, wz, W(fmic), W(tmp), W(search), W(A0), W(Ai), W(Aj), W(Ak),
                 wgl, wgl + dom.nown, p_sc, p_red1, st)
        g1 = dom.comm.allgather_dev(red)
        # alpha (conjugategrad.cpp:252-254); r -= alpha tmp with max|r| fused in; tmp = M^-1 r with dot(tmp, r) fused in (the residual
        # is zero on the ghost planes: the sum is the owned one) -- one call for the stretch up to the next gather
        lib.call("mf_cg_slab_after_dp", _ptr(g1), world, p_sc, p_state, off_w, nown, W(residual), W(tmp), p_red0,
                 sx, sy, wz, W(fmic), W(Ap), W(Aim), W(Ajm), W(Akm), p_red1, st)
        g2 = dom.comm.allgather_dev(red)
        # beta + stopping test on the device; x += alpha search and search = tmp + beta search in one pass over `search`
        lib.call("mf_cg_slab_after_zr", _ptr(g2), world, p_sc, acc32, it, p_state, off_w, nown, W(pressure), W(search), W(tmp), st)
        keep.append(g1)                                           # gathered rows stay alive until the stream has consumed them
        keep.append(g2)
        if it % STOP_POLL == 0 or it == maxIter:
            if not on_gpu or it == maxIter:
                stop, at = (int(v) for v in state.tolist())
                keep.clear()
            else:
                hstate[slot].copy_(state, non_blocking=True)
                hev[slot].record()
                stop, at = 0, 0
                if pending >= 0:
                    hev[pending].synchronize()
                    stop, at = (int(v) for v in hstate[pending].tolist())
                    del keep[:2 * STOP_POLL]                      # rows of iterations the stream has certainly consumed
                pending, slot = slot, slot ^ 1
            if stop:
                iters = at
                break
            iters = it
    if on_gpu and not stop:
        stop, at = (int(v) for v in state.tolist())               # the final state, after everything that was queued
        if stop:
            iters = at
    resNorm = float(sc[4])
    lib.call("mf_mic_check", st)
    if stop == 2:
        raise RuntimeError("GridCg::iterate: The CG solver diverged, residual norm > 1e30, stopping.")
    if stats is not None:
        stats["iterations"], stats["residual"] = iters, float(resNorm)
        stats["mic_blocking"] = (jblock, xblock)     # rows along y / cells along x of a preconditioner block (0: not cut)
    # knReplaceClampedGhostFluidVels reads the CORRECTED z-velocity of the planes below and above: the first ghost plane must come
    # out right as well, and its correction reads the pressure of the second one
    dom.exchange(pressure, 1 if phi is None else 2)
    lib.call("mf_correct_velocity", sx, sy, sz, flags.ptr, vel.ptr, pressure.ptr, st)
    if phi is not None:
        # knCorrectVelocityGhostFluid + knReplaceClampedGhostFluidVels, pressure.cpp:154-214 (read pressure / phi / flags at +-1)
        lib.call("mf_correct_velocity_ghost_fluid", sx, sy, sz, vel.ptr, flags.ptr, pressure.ptr, phi.ptr, float(gfClamp), None, 0.0, st)
        lib.call("mf_replace_clamped_ghost_fluid_vels", sx, sy, sz, vel.ptr, flags.ptr, pressure.ptr, phi.ptr, float(gfClamp), st)
    return iters


def setWallBcs(dom, flags, vel):
    s = dom.solver
    s.lib.call("mf_set_wall_bcs", dom.NX, dom.NY, dom.LZ, flags.ptr, vel.ptr, None, s.stream)


# =========================================================================================================
# FLIP on slabs: particles live on the rank that owns the plane floor(z) of their cell, in GLOBAL coordinates
# (mf_set_slab_window makes every interpolator / bounds test of the particle kernels work in global z).
#   advectInGrid      : velocity ghosts to the advection ghost width, local RK step, then migration of the particles that left
#                       the owned planes to the z-neighbour (one variable-size p2p message per side and direction)
#   mapPartsToMAC     : local scatter (incl. one ghost plane per side) -> REVERSE halo: the ghost-plane sums of vel and weight
#                       are added into the neighbour's boundary plane -> stomp / safeDivide / velOld copy
#   mapMACToParts / flipVelocityUpdate : one-plane ghosts of vel (and velOld), then the local gather
# =========================================================================================================
def slab_ranges(NZ, P):
    base, rem = divmod(NZ, P)
    z0 = [r * base + min(r, rem) for r in range(P)]
    return [(z0[r], z0[r] + base + (1 if r < rem else 0)) for r in range(P)]


class SlabParticles(object):
    """BasicParticleSystem of one slab + its pdata; every pdata created through `create` migrates with its particle"""

    def __init__(self, dom):
        self.dom = dom
        self.pp = core.BasicParticleSystem(dom.solver)

    def create(self, type):
        return self.pp.create(type)

    @property
    def np(self): return self.pp.np

    # ---- packing: a particle = 3 position floats + flag + the components of every pdata, as 4-byte columns -------------
    def _columns(self):
        pp = self.pp
        cols = [pp.pos[c * pp.cap:c * pp.cap + pp.np] for c in range(3)] + [pp.flag[:pp.np].view(torch.float32)]
        for pd in pp.pdata:
            for c in range(pd._ncomp):
                col = pd.data[c * pd.cap:c * pd.cap + pp.np]
                cols.append(col if col.dtype == torch.float32 else col.view(torch.float32))
        return cols

    def _assign(self, packed):
        """packed: [ncol][n] float32 (bit patterns)"""
        pp = self.pp
        n = packed.shape[1]
        pp.resizeAll(n, cap=max(pp.cap, n + n // 8 + 16))
        for c in range(3):
            pp.pos[c * pp.cap:c * pp.cap + n] = packed[c]
        pp.flag[:n] = packed[3].view(torch.int32)
        q = 4
        for pd in pp.pdata:
            for c in range(pd._ncomp):
                dst = pd.data[c * pd.cap:c * pd.cap + n]
                dst.copy_(packed[q] if dst.dtype == torch.float32 else packed[q].view(dst.dtype))
                q += 1

    def scatter_global(self, pos, pflag, pdata=()):
        """pos [3][n], pflag [n], pdata: list of (pdata object, array [ncomp][n] or [n]); keeps this rank's particles in order"""
        dom = self.dom
        k = np.floor(pos[2]).astype(np.int64)
        lo = -(1 << 60) if dom.below is None else dom.z0
        hi = (1 << 60) if dom.above is None else dom.z1
        m = (k >= lo) & (k < hi)
        self.pp.set_positions(np.ascontiguousarray(pos[:, m].T), pflag[m])
        for pd, arr in pdata:
            a = np.asarray(arr)
            a = a[None] if a.ndim == 1 else a
            for c in range(pd._ncomp):
                pd.data[c * pd.cap:c * pd.cap + self.pp.np] = torch.from_numpy(np.ascontiguousarray(a[c][m])).to(pd.data.device)
        return m

    def gather(self):
        """this rank's particles as numpy: pos [3][n], flag [n], one array per pdata"""
        cols = [c.detach().cpu().numpy().copy() for c in self._columns()]
        out = {"pos": np.stack(cols[:3]), "flag": cols[3].view(np.int32)}
        q = 4
        for i, pd in enumerate(self.pp.pdata):
            a = np.stack(cols[q:q + pd._ncomp])
            q += pd._ncomp
            out["pdata%d" % i] = a if pd.data.dtype == torch.float32 else a.view(np.int32)
        return out

    def migrate(self):
        """hand the particles whose cell plane is no longer owned to the z-neighbour; survivors keep their order, arrivals
        are appended (from below first).  A particle may cross one slab face per step (checked)."""
        dom, pp = self.dom, self.pp
        if dom.comm.world == 1:
            return 0
        s = dom.solver
        s.sync()
        cols = self._columns()
        packed = torch.stack(cols) if pp.np > 0 else torch.zeros((len(cols), 0), dtype=torch.float32, device=s.device)
        k = torch.floor(packed[2]).to(torch.int64)
        down = (k < dom.z0) if dom.below is not None else torch.zeros_like(k, dtype=torch.bool)
        up = (k >= dom.z1) if dom.above is not None else torch.zeros_like(k, dtype=torch.bool)
        ranges = slab_ranges(dom.NZ, dom.comm.world)
        if bool(down.any()) and int(k[down].min()) < ranges[dom.comm.rank - 1][0] and dom.comm.rank - 1 > 0:
            raise RuntimeError("slab FLIP: a particle crossed more than one slab in a step")
        if bool(up.any()) and int(k[up].max()) >= ranges[dom.comm.rank + 1][1] and dom.comm.rank + 1 < dom.comm.world - 1:
            raise RuntimeError("slab FLIP: a particle crossed more than one slab in a step")
        send_dn, send_up, stay = packed[:, down].contiguous(), packed[:, up].contiguous(), packed[:, ~(down | up)]
        from_dn, from_up = exchange_columns(dom, send_dn, send_up)
        moved = send_dn.shape[1] + send_up.shape[1] + from_dn.shape[1] + from_up.shape[1]
        if moved:
            self._assign(torch.cat([stay, from_dn, from_up], dim=1))
        return moved


def exchange_columns(dom, send_dn, send_up):
    """variable-size exchange with the two z-neighbours: [ncol][n] float32 blocks (bit patterns) go down / up, the blocks the
    neighbours sent come back as (from_below, from_above).  One count message + one payload message per side and direction."""
    s = dom.solver
    cnt = torch.tensor([send_dn.shape[1], send_up.shape[1]], dtype=torch.int64, device=s.device)
    got = torch.zeros(2, dtype=torch.int64, device=s.device)
    pairs = []
    if dom.below is not None:
        pairs.append((cnt[0:1], got[0:1], dom.below))
    if dom.above is not None:
        pairs.append((cnt[1:2], got[1:2], dom.above))
    dom.comm.sendrecv(pairs)
    # payloads (only the non-empty ones; both sides know the counts)
    ncol = send_dn.shape[0]
    from_dn = torch.zeros((ncol, int(got[0])), dtype=torch.float32, device=s.device)
    from_up = torch.zeros((ncol, int(got[1])), dtype=torch.float32, device=s.device)
    pairs = []
    if dom.below is not None and (send_dn.shape[1] or from_dn.shape[1]):
        pairs.append((send_dn if send_dn.shape[1] else None, from_dn if from_dn.shape[1] else None, dom.below))
    if dom.above is not None and (send_up.shape[1] or from_up.shape[1]):
        pairs.append((send_up if send_up.shape[1] else None, from_up if from_up.shape[1] else None, dom.above))
    dom.comm.sendrecv(pairs)
    return from_dn, from_up


def reduce_ghosts(dom, grid, width=1):
    """reverse halo: add my ghost-plane values into the neighbour's boundary planes (and receive theirs into mine)"""
    if dom.comm.world == 1:
        return
    w = int(width)
    top = dom.gl + dom.nown
    pairs, adds = [], []
    if dom.below is not None:
        tmp = torch.empty_like(dom.planes(grid, dom.gl, dom.gl + w).contiguous())
        pairs.append((dom.planes(grid, dom.gl - w, dom.gl), tmp, dom.below))
        adds.append((dom.planes(grid, dom.gl, dom.gl + w), tmp))
    if dom.above is not None:
        tmp = torch.empty_like(dom.planes(grid, top - w, top).contiguous())
        pairs.append((dom.planes(grid, top, top + w), tmp, dom.above))
        adds.append((dom.planes(grid, top - w, top), tmp))
    dom.solver.sync()
    dom.comm.sendrecv(pairs)
    for dst, tmp in adds:
        dst.add_(tmp)


def advectInGrid(dom, sp, flags, vel, integrationMode, deleteInObstacle=True, stopInObstacle=True, skipNew=False, ptype=None,
                 exclude=0, migrate=True):
    """ParticleSystem::advectInGrid on a slab (particle.h:526-550): ghosts of `vel` to the ghost width, local step, migration
    (migrate=False: the caller moves the particles further -- eulerStep, projectOutOfBnd, pushOutofObs -- and calls sp.migrate()
    itself once the positions of the step are final)"""
    s = dom.solver
    if dom.comm.world > 1:
        m = 0.0
        for c in range(3):
            m = max(m, dom.max_abs_owned(vel, comp=c))
        reach = int(math.ceil(m * s.getDt())) + 2          # furthest sub-step position + trilinear/MAC-shift reach
        if reach > dom.G:
            raise RuntimeError("slab advectInGrid: max|v| dt = %.2f needs %d ghost planes, domain has %d" % (m * s.getDt(), reach, dom.G))
    dom.exchange(vel)
    sp.pp.advectInGrid(flags, vel, integrationMode, deleteInObstacle=deleteInObstacle, stopInObstacle=stopInObstacle, skipNew=skipNew,
                       ptype=ptype, exclude=exclude)
    return sp.migrate() if migrate else 0


def mapPartsToMAC(dom, flags, vel, velOld, sp, partVel, weight=None, deterministic=True, ptype=None, exclude=0):
    """mapPartsToMAC (flip.cpp:637-661) on a slab: local scatter, reverse halo of the pre-division sums, then the division"""
    s = dom.solver
    w = weight if weight is not None else core.VecGrid(s)
    pp = sp.pp
    s.lib.call("mf_map_parts_to_mac_accum", dom.NX, dom.NY, dom.LZ, vel.ptr, w.ptr, pp.np, pp.cap, _ptr(pp.pos), _ptr(pp.flag),
               partVel.ptr, None if ptype is None else ptype.ptr, int(exclude), int(deterministic), s.stream)
    reduce_ghosts(dom, vel, 1)
    reduce_ghosts(dom, w, 1)
    s.lib.call("mf_map_parts_to_mac_finish", 3 * vel.n, vel.ptr, velOld.ptr, w.ptr, s.stream)


def markFluidCells(dom, sp, flags, ptype=None, exclude=0):
    """markFluidCells (flip.cpp:166-188) on a slab: every rank marks the cells of its own particles, ghost flags come from
    their owners"""
    from . import plugins
    plugins.markFluidCells(sp.pp, flags, ptype=ptype, exclude=exclude)
    dom.exchange(flags)


def _need_ghost(dom, passes, what):
    if dom.comm.world > 1 and passes + 1 > dom.G:
        raise RuntimeError("slab %s: %d passes need %d ghost planes, domain has %d" % (what, passes, passes + 1, dom.G))


def extrapolateMACFromWeight(dom, vel, weight, distance=2):
    """fastmarch.cpp:415-430: `distance` one-cell passes -> run on ghosts `distance`+1 deep, the owned planes come out exact"""
    from . import plugins
    _need_ghost(dom, distance, "extrapolateMACFromWeight")
    dom.exchange(vel)
    dom.exchange(weight)
    plugins.extrapolateMACFromWeight(vel, weight, distance)


def extrapolateMACSimple(dom, flags, vel, distance=4):
    """fastmarch.cpp:337-376, same argument"""
    from . import plugins
    _need_ghost(dom, distance, "extrapolateMACSimple")
    dom.exchange(vel)
    plugins.extrapolateMACSimple(flags, vel, distance)


def addGravity(dom, flags, vel, gravity):
    from . import plugins
    plugins.addGravity(flags, vel, gravity)


def mapMACToParts(dom, flags, vel, sp, partVel):
    s, pp = dom.solver, sp.pp
    dom.exchange(vel, 1)
    s.lib.call("mf_map_mac_to_parts", dom.NX, dom.NY, dom.LZ, vel.ptr, pp.np, pp.cap, _ptr(pp.pos), _ptr(pp.flag), partVel.ptr, None, 0, s.stream)


def flipVelocityUpdate(dom, flags, vel, velOld, sp, partVel, flipRatio, ptype=None, exclude=0):
    s, pp = dom.solver, sp.pp
    dom.exchange(vel, 1)
    dom.exchange(velOld, 1)
    s.lib.call("mf_flip_velocity_update", dom.NX, dom.NY, dom.LZ, vel.ptr, velOld.ptr, pp.np, pp.cap, _ptr(pp.pos), _ptr(pp.flag),
               partVel.ptr, float(flipRatio), None if ptype is None else ptype.ptr, int(exclude), s.stream)


# =========================================================================================================
# free-surface pieces of scenes/benchmark_dam.py on slabs (BASELINE config 4): the ghost-fluid pressure solve needs the particle
# level set, which reaches across the slab faces in two ways -- a particle's sphere covers cells of the neighbour slab
# (unionParticleLevelset looks one cell around every cell), and extrapolateLsSimple walks `distance` cells away from the surface.
# =========================================================================================================
def max_abs_mac(dom, vel):
    """MACGrid::getMaxAbs over the whole domain (sqrt of the largest |v|^2, grid.cpp:364-366): local maxima over the owned
    planes, one all-gather, max -- order independent, so bit-identical to the undivided grid"""
    s = dom.solver
    own = dom.planes(vel, dom.gl, dom.gl + dom.nown).contiguous()
    r = ctypes.c_float()
    s.lib.call("mf_grid_max_abs_vec3", dom.n_own, _ptr(own), ctypes.byref(r), s.stream)
    g = dom.comm.gather_scalars([r.value], s.device)
    return float(np.max(g[:, 0]))


def adaptTimestep(dom, vel):
    """s.adaptTimestep(vel.getMaxAbs()) (fluidsolver.cpp:184-204) with the domain-wide maximum: every rank takes the same dt"""
    dom.solver.adaptTimestep(max_abs_mac(dom, vel))


def halo_particles(dom, sp, width, ptype=None):
    """A particle system holding this rank's particles followed by the neighbours' particles whose cell lies within `width`
    planes of the slab faces (positions, flags and -- if given -- the ptype column travel; arrivals from below first).  Used
    where a cell's value depends on the particles of the cells around it."""
    s, pp = dom.solver, sp.pp
    halo = core.BasicParticleSystem(s)
    hpt = halo.create(core.PdataInt) if ptype is not None else None
    n = pp.np
    cols = [pp.pos[c * pp.cap:c * pp.cap + n] for c in range(3)] + [pp.flag[:n].view(torch.float32)]
    if ptype is not None:
        cols.append(ptype.data[:n].view(torch.float32))
    packed = torch.stack(cols) if n > 0 else torch.zeros((len(cols), 0), dtype=torch.float32, device=s.device)
    if dom.comm.world > 1:
        s.sync()
        k = torch.floor(packed[2]).to(torch.int64)
        dn = (k < dom.z0 + width) if dom.below is not None else torch.zeros_like(k, dtype=torch.bool)
        up = (k >= dom.z1 - width) if dom.above is not None else torch.zeros_like(k, dtype=torch.bool)
        from_dn, from_up = exchange_columns(dom, packed[:, dn].contiguous(), packed[:, up].contiguous())
        packed = torch.cat([packed, from_dn, from_up], dim=1)
    m = packed.shape[1]
    halo.resizeAll(m)
    if m:
        for c in range(3):
            halo.pos[c * halo.cap:c * halo.cap + m] = packed[c]
        halo.flag[:m] = packed[3].view(torch.int32)
        if hpt is not None:
            hpt.data[:m] = packed[4].view(torch.int32)
    return halo, hpt


def unionParticleLevelset(dom, sp, flags, phi, indexSys=None, index=None, radiusFactor=1.0, ptype=None, exclude=0):
    """gridParticleIndex + unionParticleLevelset (flip.cpp:273-363) on a slab.  phi of a cell is the minimum over the particles of
    the cells within r = int(radius) + 1 planes (1 for radiusFactor 1), so the index is built over this rank's particles plus
    the neighbours' particles of the r planes beyond each slab face: the owned planes come out bit-identical to the undivided
    domain (a minimum does not depend on the order); the ghost planes are then fetched from their owners."""
    from . import plugins
    s = dom.solver
    radius = 0.5 * math.sqrt(3.0) * (float(radiusFactor) + 0.01)            # calculateRadiusFactor, flip.cpp:198-200
    r = int(radius) + 1
    if dom.comm.world > 1 and r > dom.G:
        raise RuntimeError("slab unionParticleLevelset: particle radius reaches %d planes, domain has %d ghost planes" % (r, dom.G))
    halo, hpt = halo_particles(dom, sp, r, ptype)
    indexSys = indexSys if indexSys is not None else core.ParticleIndexSystem(s)
    index = index if index is not None else core.IntGrid(s)
    plugins.gridParticleIndex(halo, indexSys, flags, index)
    plugins.unionParticleLevelset(halo, indexSys, flags, index, phi, radiusFactor, ptype=hpt, exclude=exclude)
    dom.exchange(phi)


def extrapolateLsSimple(dom, phi, distance=4, inside=False, include_walls=False):
    """fastmarch.cpp:472-522: a mark pass, a first layer and `distance` - 1 one-cell passes.  Run on ghosts `distance` + 2 deep
    (phi ghosts must be current): every pass spoils one more plane from the outside, the owned planes come out exact."""
    from . import plugins
    _need_ghost(dom, distance + 1, "extrapolateLsSimple")
    plugins.extrapolateLsSimple(phi, distance, inside, include_walls)


def markIsolatedFluidCell(dom, flags, mark):
    """grid.cpp:987-1011 (reads the six neighbours' flags): flag ghosts must be current; the outermost ghost plane has no outer
    neighbour here, so the ghosts are fetched again afterwards"""
    from . import plugins
    plugins.markIsolatedFluidCell(flags, mark)
    dom.exchange(flags)


# =========================================================================================================
# the up-res loop of scenes/waveletTurbulence.py on slabs (BASELINE config 5): a coarse solver `sm` and an `upres` times finer
# solver `xl` on the same z-ranges (SlabDomain + refine).  Everything except the pressure solve is bit-identical to the
# undivided domain: the operators below only fetch the ghosts their stencil reaches.
# =========================================================================================================
def vorticityConfinement(dom, vel, flags, strength=0.):
    """extforces.cpp:409-428: velCenter (reads vel at +1), curl (+-1), |curl|, its gradient (+-1) -> three planes of reach;
    run on the full ghost width, the owned planes come out exact"""
    from . import plugins
    if dom.comm.world > 1 and dom.G < 4:
        raise RuntimeError("slab vorticityConfinement needs 4 ghost planes, domain has %d" % dom.G)
    dom.exchange(vel)
    plugins.vorticityConfinement(vel, flags, strength)


def addBuoyancy(dom, flags, density, vel, gravity):
    """extforces.cpp:73-88 (reads density at -1): density ghosts must be current"""
    from . import plugins
    plugins.addBuoyancy(flags, density, vel, gravity)


def gather_window(dom, grid, w0, w1):
    """planes [w0, w1) of a Real grid of the whole domain as one contiguous device tensor [w1 - w0][NY * NX], fetched from
    their owners (any number of ranks away; every rank calls this with its own window)"""
    s, comm = dom.solver, dom.comm
    out = torch.empty((w1 - w0, dom.XY), dtype=torch.float32, device=s.device)
    mine = dom.planes(grid, dom.gl, dom.gl + dom.nown)[0]
    a, b = max(w0, dom.z0), min(w1, dom.z1)
    if a < b:
        out[a - w0:b - w0] = mine[a - dom.z0:b - dom.z0]
    if comm.world > 1:
        # every rank's window, so that owners know what to send
        wins = comm.gather_scalars([float(w0), float(w1)], s.device).astype(np.int64)
        ranges = slab_ranges_of(dom)
        s.sync()
        pairs = []
        for q in range(comm.world):
            if q == comm.rank:
                continue
            qa, qb = max(int(wins[q][0]), dom.z0), min(int(wins[q][1]), dom.z1)        # my planes inside q's window
            ra, rb = max(w0, ranges[q][0]), min(w1, ranges[q][1])                        # q's planes inside my window
            snd = mine[qa - dom.z0:qb - dom.z0] if qa < qb else None
            rcv = out[ra - w0:rb - w0] if ra < rb else None
            if snd is not None or rcv is not None:
                pairs.append((snd, rcv, q))
        comm.sendrecv(pairs)
    return out


def slab_ranges_of(dom):
    """owned z-range of every rank (the even split, or the refined one)"""
    g = dom.comm.gather_scalars([float(dom.z0), float(dom.z1)], dom.solver.device).astype(np.int64)
    return [(int(a), int(b)) for a, b in g]


WAVELET_REACH = 20     # planes: upsample reads coarse samples i/2-1 .. i/2+2, each the taps 2k-16 .. 2k+15, + 1 for the smoothing


def computeWaveletCoeffs(dom, energy):
    """computeWaveletCoeffs (waveletturbulence.cpp:197-201 -> WaveletNoiseField::computeCoefficients, noisefield.cpp:233-300) on a
    slab.  The decomposition filters whole lines along x, y and z; a value depends on the 20 planes either side of it.  Every rank
    therefore runs the filter on a WINDOW of planes [z0 - 20, z1 + 20) (clipped to the domain, even start: the down-sampling pairs
    planes 2m, 2m+1) gathered from their owners and keeps its own planes: same taps in the same order as the undivided filter,
    bit-identical.  The window is not a solver grid (its depth differs), so the library is called on raw tensors."""
    s = dom.solver
    if dom.comm.world == 1:
        from . import plugins
        return plugins.computeWaveletCoeffs(energy)
    w0 = max(0, (dom.z0 - WAVELET_REACH) & ~1)
    w1 = min(dom.NZ, dom.z1 + WAVELET_REACH)
    win = gather_window(dom, energy, w0, w1)
    t1, t2 = torch.empty_like(win), torch.empty_like(win)
    s.lib.call("mf_compute_wavelet_coeffs", dom.NX, dom.NY, w1 - w0, _ptr(win), _ptr(t1), _ptr(t2), s.stream)
    dom.planes(energy, dom.gl, dom.gl + dom.nown)[0].copy_(win[dom.z0 - w0:dom.z1 - w0])


def interpolateGrid(cdom, fdom, target, source):
    """interpolateGrid (waveletturbulence.cpp:37-56) from the coarse slab to the fine slab: a fine cell samples the coarse
    grid around its own (global) position, i.e. the coarse planes of its own rank plus at most one ghost plane -- run on every
    fine plane, exact wherever the coarse ghosts reach (source ghosts must be current)"""
    from . import plugins
    plugins.interpolateGrid(target=target, source=source)


def interpolateMACGrid(cdom, fdom, target, source):
    from . import plugins
    plugins.interpolateMACGrid(target=target, source=source)


# =========================================================================================================
# bench.py --gpus N>1
# =========================================================================================================
def global_flags(n):
    f = np.full((n, n, n), 1, np.int32)
    f[:, :, 0] = f[:, :, -1] = f[:, 0, :] = f[:, -1, :] = 2
    f[0] = f[-1] = 2
    return f


def smoke_step(dom, flags, vel, vel0, dens, pres, stats):
    """the bench step on a slab: same operator sequence as the single-GPU step in bench.py"""
    vel.copyFrom(vel0)
    dom.exchange(dens)
    advectSemiLagrange(dom, flags, vel, dens, order=2)
    advectSemiLagrange(dom, flags, vel, vel, order=2)
    dom.exchange(vel, 1)
    setWallBcs(dom, flags, vel)
    solvePressure(dom, vel, pres, flags, stats=stats)


def bench_slab_step(n, dt, steps, warmup, rank, world):
    import time
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    G = required_ghost(2.0)
    dom = SlabDomain((n, n, n), G)
    s = dom.solver
    s.timestep = dt
    flags, vel, vel0, dens, pres = core.FlagGrid(s), core.MACGrid(s), core.MACGrid(s), core.Grid(s), core.Grid(s)
    dom.scatter_global(flags, global_flags(n))
    v = bench.synthetic_velocity(n, n, n)
    dom.scatter_global(vel0, v)
    del v
    setWallBcs(dom, flags, vel0)
    dom.exchange(vel0)
    dom.scatter_global(dens, bench.synthetic_density(n, n, n))
    stats, iters = {}, []
    for _ in range(warmup):
        smoke_step(dom, flags, vel, vel0, dens, pres, stats)
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.freeze()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        smoke_step(dom, flags, vel, vel0, dens, pres, stats)
        iters.append(stats["iterations"])
    torch.cuda.synchronize()
    dist.barrier()
    el = time.perf_counter() - t0
    return {"elapsed": el, "cells": n ** 3, "cg_iterations": iters,
            "notes": "z-slab x%d: ghost width %d planes, slab-local MIC(0) (block-Jacobi across slab faces), "
                     "1-plane p2p halo + all-gather of CG scalars per iteration" % (world, G)}

"""-m gpu: the HIP product path (through the C ABI) against
  (1) the committed golden vectors produced by the compiled reference,
  (2) the oracle restatement on the same seeded inputs (and the compiled reference itself when its .so travelled),
  (3) size-independent properties at BASELINE.json's 256^3.
Bars: bit-exact for flag/index work and for every kernel without a reduction; <= 1e-5 relative (stated per test)
where fp64 partial sums are combined in a different order than the reference's thread-local sums (CG scalars) or
where fp32 atomics are used (non-deterministic particle->grid mode)."""
import ctypes

import numpy as np
import pytest
import torch

import cases
import util
from util import assert_bitexact, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5   # BASELINE.json north_star: "within 1e-5 relative for fp32 grid fields"


def _close(a, b, what, tol=TOL):
    e = rel_err(a, b)
    assert e <= tol, "%s: relative error %g > %g" % (what, e, tol)


def test_native_library_is_the_one_loaded(hip, hip_backend):
    from mantaflow_amd import _lib
    lib = _lib.get()
    assert lib.backend == "hip" and lib.path.endswith("libmanta_hip.so")
    assert hip.lib.backend == "hip"
    maps = open("/proc/self/maps").read()
    assert "libmanta_hip.so" in maps


def test_golden_vectors(hip, hip_backend):
    gold = cases.load_golden()
    got = cases.golden_outputs(hip, deterministic_p2g=True)
    assert set(got) == set(gold)
    exact, close = [], []
    for k in sorted(gold):
        if k.startswith("cgstat"):
            assert got[k][0] == gold[k][0], "%s: iteration count %s vs %s" % (k, got[k][0], gold[k][0])
            _close(got[k][1:], gold[k][1:], k, 1e-4)
            continue
        reduction_dependent = (k.startswith(("cg_", "sp_")) and not k.endswith("_rhs")) or k.startswith(("plume_", "wlt2d_", "wlt3d_"))
        if reduction_dependent:
            _close(got[k], gold[k], k)
            (exact if np.array_equal(got[k], gold[k]) else close).append(k)
        else:
            assert_bitexact(got[k], gold[k], k)
            exact.append(k)
    print("bit-exact: %d arrays; within %g: %s" % (len(exact), TOL, close))


def test_golden_p2g_atomic_mode(hip, hip_backend):
    """opt-in atomic particle->grid mode (setDeterministicP2G(False)): order of fp32 sums is undefined -> 1e-5 relative"""
    gold = cases.load_golden()
    dims = (12, 10, 9)
    flags = util.make_flags(*dims, 14, empty_top=True)
    vel, velOld = util.rand_vel(*dims, 15), util.rand_vel(*dims, 16)
    pos, pflag, pvel = util.make_particles(flags, 3, 17)
    r = cases.run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel, deterministic=False)
    for k in ("p2g_vel", "p2g_velOld", "p2g_weight", "p2g_real", "p2g_vec3"):
        _close(r[k], gold["flip_" + k], k)
    for k in ("pic", "flip", "g2p_real", "g2p_vec3"):
        assert_bitexact(r[k], gold["flip_" + k], k)


# ---- kernel-level: HIP vs oracle on the CPU-test scenarios ------------------------------------------------
DIMS = cases.SIZES_3D + [cases.SIZE_2D, (32, 24, 40)]


@pytest.mark.parametrize("dims", DIMS)
def test_laplace_and_apply_matrix(hip, oracle, dims):
    flags, fr = cases.laplace_inputs(dims, 3, True)
    for f in (None, fr):
        for x, y in zip(cases.run_laplace_impl(hip, dims, flags, f), cases.run_laplace_impl(oracle, dims, flags, f)):
            assert_bitexact(x, y, "MakeLaplaceMatrix")
    for seed in (1, 2):
        flags, A, src = cases.system_inputs(dims, seed)
        assert_bitexact(cases.run_apply_matrix_impl(hip, dims, flags, A, src), cases.run_apply_matrix_impl(oracle, dims, flags, A, src), "ApplyMatrix")


@pytest.mark.parametrize("dims", cases.SIZES_3D + [(32, 24, 40), (9, 9, 9), (64, 8, 8)])
def test_mic(hip, oracle, dims):
    for seed in (1, 2):
        flags, A, src = cases.system_inputs(dims, seed)
        ap, dst = cases.run_mic_impl(hip, dims, flags, A, src)
        ap_o, dst_o = cases.run_mic_impl(oracle, dims, flags, A, src)
        assert_bitexact(ap, ap_o, "Aprecond")
        assert_bitexact(dst, dst_o, "mic apply")


@pytest.mark.parametrize("dims", [(64, 16, 16), (128, 24, 16), (40, 17, 12)])
def test_mic_apply_signed_zeros_negative_and_huge_operands(hip, oracle, dims):
    """The packed path of the row sweeps keeps A * Aprecond in its operand ring and forms x * (A * p) where the reference forms
    (x * A) * p -- equal bit for bit for A in {+0, -1} whatever x and p are.  Checked here on operands the solver never produces:
    Aprecond with negative, zero, -0.0 and denormal entries, a right-hand side with signed zeros, denormals and values large enough to
    overflow to +-inf (and then NaN) inside the sweep.  Non-NaN results must agree bit for bit, NaNs must sit in the same cells."""
    sx, sy, sz = dims
    flags, A, src = cases.system_inputs(dims, 7)
    rng = np.random.default_rng(99)
    src = src.copy()
    src[rng.random(src.shape) < 0.15] = 0.0
    src[rng.random(src.shape) < 0.10] = -0.0
    src[rng.random(src.shape) < 0.03] = np.float32(1e-41)
    src[rng.random(src.shape) < 0.002] = np.float32(3e37)
    res = []
    for impl in (hip, oracle):
        f = impl.dev(flags)
        dA = [impl.dev(a) for a in A]
        ap = impl.dev(np.zeros((sz, sy, sx), np.float32))
        impl.call("mf_mic_init", sx, sy, sz, f, ap, dA[0], dA[1], dA[2], dA[3], None)
        impl.sync()
        ap_h = impl.host(ap).copy()
        r2 = np.random.default_rng(5)
        ap_h[r2.random(ap_h.shape) < 0.10] *= np.float32(-1.0)
        ap_h[r2.random(ap_h.shape) < 0.05] = 0.0
        ap_h[r2.random(ap_h.shape) < 0.05] = -0.0
        ap_h[r2.random(ap_h.shape) < 0.02] = np.float32(1e-40)
        ap2 = impl.dev(ap_h)
        dst = impl.dev(np.full((sz, sy, sx), 0.25, np.float32))
        impl.call("mf_mic_apply", sx, sy, sz, f, dst, impl.dev(src), ap2, dA[1], dA[2], dA[3], None)
        impl.sync()
        res.append(impl.host(dst))
    got, want = res
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), "NaNs in different cells: %d vs %d" % (nan_g.sum(), nan_w.sum())
    assert (~nan_w).sum() > 0.2 * want.size
    assert np.array_equal(got[~nan_w].view(np.uint32), want[~nan_w].view(np.uint32)), "finite results differ in %d cells" % (
        (got[~nan_w].view(np.uint32) != want[~nan_w].view(np.uint32)).sum())


@pytest.mark.parametrize("mode", ["rows", "levels"])
@pytest.mark.parametrize("dims", [(32, 24, 40), (37, 21, 19), (64, 64, 64), (24, 40, 9), (16, 8, 136), (40, 33, 27), (128, 40, 24), (32, 8, 8),
                                  (96, 72, 17)])
def test_mic_every_sweep_mode(hip, oracle, dims, mode):
    """the ways the MIC sweeps are parallelised (mf_set_mic_mode) give the serial sweep's bits, and a CG solve takes the same
    number of iterations in each: odd bundle counts, rows outside the grid and a single bundle are among the sizes.  The mode is
    taken at mf_mic_init and stays with the system it registers (an unknown name is refused)."""
    assert hip.lib.cdll.mf_set_mic_mode(b"tiles") != 0
    flags, A, src = cases.system_inputs(dims, 3)
    rhs = cases.cg_rhs(dims, flags, 3)
    ap_o, dst_o = cases.run_mic_impl(oracle, dims, flags, A, src)
    xo, sto = cases.run_cg_impl(oracle, dims, flags, A, rhs, 2, 1e-4, 50, 0)
    assert hip.lib.cdll.mf_set_mic_mode(mode.encode()) == 0
    try:
        ap, dst = cases.run_mic_impl(hip, dims, flags, A, src)
        x, st = cases.run_cg_impl(hip, dims, flags, A, rhs, 2, 1e-4, 50, 0)
    finally:
        assert hip.lib.cdll.mf_set_mic_mode(None) == 0
    assert_bitexact(ap, ap_o, "Aprecond " + mode)
    assert_bitexact(dst, dst_o, "mic apply " + mode)
    assert st[0] == sto[0], (mode, st, sto)
    _close(x, xo, "cg solution " + mode)


@pytest.mark.parametrize("a0", [15.0, 16.0, 2.5, -0.0, 0.0])
def test_apply_matrix_packed_diagonal_nibble(hip, oracle, a0):
    """ApplyMatrix inside the PCG reads the diagonal from bits 4-7 of the packed byte when every fluid cell's A0 is a small
    non-negative integer (k_mic_pack's ok[1], bit patterns compared).  Systems whose A0 holds 15 (largest code), 16, 2.5 or -0.0 in
    some fluid cells: whichever form is taken (nibble, or the A0 array where the nibble would not be exact), the result has the
    bits of the reference expression -- signed zeros included."""
    dims = (32, 24, 16)
    sx, sy, sz = dims
    flags, A, src = cases.system_inputs(dims, 9)
    A = [a.copy() for a in A]
    fluid = np.argwhere((flags & util.FLUID) != 0)
    rng = np.random.default_rng(31)
    for k, j, i in fluid[rng.choice(len(fluid), 40, replace=False)]:
        A[0][k, j, i] = np.float32(a0)
    src = src.copy()
    src[rng.random(src.shape) < 0.3] = 0.0          # zero sums, so that the sign of a rebuilt zero diagonal would show
    src[rng.random(src.shape) < 0.2] = -0.0
    want = cases.run_apply_matrix_impl(oracle, dims, flags, A, src)
    f, s_, dA = hip.dev(flags), hip.dev(src), [hip.dev(a) for a in A]
    ap = hip.dev(np.zeros((sz, sy, sx), np.float32))
    dst = hip.dev(np.full((sz, sy, sx), 7.0, np.float32))
    hip.call("mf_mic_init", sx, sy, sz, f, ap, *dA, None)
    us = ctypes.c_double()
    hip.call("mf_time_apply_matrix_packed", sx, sy, sz, f, dst, s_, *dA, 1, ctypes.byref(us), None)
    hip.sync()
    got = hip.host(dst)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "packed ApplyMatrix differs in %d cells (A0 = %r)" % (
        (got.view(np.uint32) != want.view(np.uint32)).sum(), a0)


@pytest.mark.parametrize("dims", [(32, 24, 40), (64, 48, 24), (36, 20, 17)])
def test_apply_matrix_with_packed_coefficients(hip, oracle, dims):
    """mf_pack_matrix: mf_apply_matrix on the packed {fluid, Ai, Aj, Ak} bytes gives the bits of the general kernel and of the
    oracle; a matrix that is not all 0 / -1 is refused (the general kernel keeps running), other pointers are not affected"""
    sx, sy, sz = dims
    flags, A, src = cases.system_inputs(dims, 6)
    want = cases.run_apply_matrix_impl(oracle, dims, flags, A, src)
    f, s_, dA = hip.dev(flags), hip.dev(src), [hip.dev(a) for a in A]
    dst = hip.dev(np.full((sz, sy, sx), 7.0, np.float32))
    hip.call("mf_apply_matrix", sx, sy, sz, f, dst, s_, *dA, None)
    hip.sync()
    assert_bitexact(hip.host(dst), want, "general kernel")
    hip.call("mf_pack_matrix", sx, sy, sz, f, dA[0], dA[1], dA[2], dA[3], None)
    dst2 = hip.dev(np.full((sz, sy, sx), 7.0, np.float32))
    hip.call("mf_apply_matrix", sx, sy, sz, f, dst2, s_, *dA, None)
    hip.sync()
    assert_bitexact(hip.host(dst2), want, "packed kernel")
    # a matrix with other values: the packed bytes are refused, the result is still right
    B = [a.copy() for a in A]
    B[2] *= np.float32(0.5)
    wantB = cases.run_apply_matrix_impl(oracle, dims, flags, B, src)
    dB = [hip.dev(b) for b in B]
    hip.call("mf_pack_matrix", sx, sy, sz, f, dB[0], dB[1], dB[2], dB[3], None)
    dst3 = hip.dev(np.full((sz, sy, sx), 7.0, np.float32))
    hip.call("mf_apply_matrix", sx, sy, sz, f, dst3, s_, *dB, None)
    hip.sync()
    assert_bitexact(hip.host(dst3), wantB, "scaled matrix after mf_pack_matrix")


@pytest.mark.parametrize("dims", [(32, 24, 40), (64, 48, 24)])
@pytest.mark.parametrize("variant", ["scaled", "one_cell", "minus_zero"])
def test_mic_and_cg_with_a_matrix_that_cannot_be_packed(hip, oracle, dims, variant):
    """the packed-operand path (flags + Ai + Aj + Ak as one byte per cell) is taken only when every off-diagonal is exactly +0
    or -1; a matrix with other values (second-order boundaries scale them by face fractions) must go through the four-array
    path and still give the serial sweep's bits -- also when a single cell is off, or when a coefficient is -0"""
    flags, A, src = cases.system_inputs(dims, 5)
    A = [a.copy() for a in A]
    if variant == "scaled":
        A[1] *= np.float32(0.75); A[2] *= np.float32(0.5); A[3] *= np.float32(0.875)
    elif variant == "one_cell":
        kk, jj, ii = np.argwhere(A[2] != 0)[len(np.argwhere(A[2] != 0)) // 2]
        A[2][kk, jj, ii] = np.float32(-0.5)
    else:
        A[3][A[3] == 0] = np.float32(-0.0)
    rhs = cases.cg_rhs(dims, flags, 5)
    ap_o, dst_o = cases.run_mic_impl(oracle, dims, flags, A, src)
    xo, sto = cases.run_cg_impl(oracle, dims, flags, A, rhs, 2, 1e-4, 40, 0)
    ap, dst = cases.run_mic_impl(hip, dims, flags, A, src)
    x, st = cases.run_cg_impl(hip, dims, flags, A, rhs, 2, 1e-4, 40, 0)
    assert_bitexact(ap, ap_o, "Aprecond")
    assert_bitexact(dst, dst_o, "mic apply")
    assert st[0] == sto[0], (st, sto)
    _close(x, xo, "cg solution")


@pytest.mark.parametrize("dims,rows", [((32, 64, 24), 16), ((24, 100, 40), 32), ((40, 72, 17), 24)])
def test_mic_blocked_sweeps_equal_serial_sweep_of_cut_system(hip, oracle, dims, rows):
    """mf_mic_init_blocked (multi-GPU block-Jacobi in y): with the Aj coupling zeroed at the block faces the row-streaming
    sweeps skip those hand-offs and still give the bits of the serial sweep over the same cut coefficients.  The blocking
    belongs to the system it was given with: an uncut system solved afterwards in the same process is the plain reference sweep."""
    sx, sy, sz = dims
    flags, A0, src = cases.system_inputs(dims, 7)
    A = [a.copy() for a in A0]
    for jc in range(rows, sy, rows):
        A[2][:, jc - 1, :] = 0          # Aj couples rows j and j+1
    ap_o, dst_o = cases.run_mic_impl(oracle, dims, flags, A, src)
    apu_o, dstu_o = cases.run_mic_impl(oracle, dims, flags, A0, src)
    ap, dst, (apu, dstu) = cases.run_mic_impl(hip, dims, flags, A, src, blocking=(rows, 0), then_plain=A0)
    ap2, dst2 = cases.run_mic_impl(hip, dims, flags, A, src, blocking=(rows, 0))
    assert_bitexact(ap, ap_o, "Aprecond (cut)")
    assert_bitexact(dst, dst_o, "blocked mic apply")
    assert_bitexact(dst, dst2, "blocked mic apply re-run")
    assert_bitexact(apu, apu_o, "Aprecond (uncut system after a blocked one)")
    assert_bitexact(dstu, dstu_o, "uncut mic apply after a blocked one")
    with pytest.raises(RuntimeError):
        cases.run_mic_impl(hip, dims, flags, A, src, blocking=(12, 0))


@pytest.mark.parametrize("dims,rows,cells", [((64, 64, 24), 16, 32), ((72, 40, 17), 0, 24), ((100, 48, 16), 24, 48), ((256, 64, 16), 32, 64)])
def test_mic_x_blocked_sweeps_equal_serial_sweep_of_cut_system(hip, oracle, dims, rows, cells):
    """mf_mic_init_blocked, x-blocks (Ai zeroed at the block faces), alone and together with the y-blocking; the last x-block may
    be shorter than the others"""
    sx, sy, sz = dims
    flags, A0, src = cases.system_inputs(dims, 9)
    A = [a.copy() for a in A0]
    for ic in range(cells, sx, cells):
        A[1][:, :, ic - 1] = 0          # Ai couples cells i and i+1
    if rows:
        for jc in range(rows, sy, rows):
            A[2][:, jc - 1, :] = 0
    ap_o, dst_o = cases.run_mic_impl(oracle, dims, flags, A, src)
    apu_o, dstu_o = cases.run_mic_impl(oracle, dims, flags, A0, src)
    ap, dst, (apu, dstu) = cases.run_mic_impl(hip, dims, flags, A, src, blocking=(rows, cells), then_plain=A0)
    ap2, dst2 = cases.run_mic_impl(hip, dims, flags, A, src, blocking=(rows, cells))
    assert_bitexact(ap, ap_o, "Aprecond (cut)")
    assert_bitexact(dst, dst_o, "x-blocked mic apply")
    assert_bitexact(dst, dst2, "x-blocked mic apply re-run")
    assert_bitexact(dstu, dstu_o, "uncut mic apply after an x-blocked one")
    with pytest.raises(RuntimeError):
        cases.run_mic_impl(hip, dims, flags, A, src, blocking=(0, 12))


def test_mic_mode_rejects_unknown_name(hip):
    assert hip.lib.cdll.mf_set_mic_mode(b"diagonal") != 0
    assert b"unknown mode" in hip.lib.cdll.mf_last_error()
    assert hip.lib.cdll.mf_set_mic_mode(None) == 0


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("pc,acc,iters,l2", [(2, 1e-3, 60, 0), (2, 1e-9, 3, 0), (0, 1e-3, 80, 0), (2, 1e-4, 60, 1)])
def test_cg_solve(hip, oracle, dims, pc, acc, iters, l2):
    flags, A, _ = cases.system_inputs(dims, 5)
    rhs = cases.cg_rhs(dims, flags, 5)
    x, st = cases.run_cg_impl(hip, dims, flags, A, rhs, pc, acc, iters, l2)
    xo, sto = cases.run_cg_impl(oracle, dims, flags, A, rhs, pc, acc, iters, l2)
    assert st[0] == sto[0], (st, sto)
    _close(x, xo, "cg solution")
    _close(st[1:], sto[1:], "resNorm/sigma", 1e-4)


def test_cg_diverged_reports_like_reference(hip):
    dims = (16, 16, 16)
    flags, A, _ = cases.system_inputs(dims, 4)     # closed box, random rhs: inconsistent system
    rhs = cases.cg_rhs(dims, flags, 4) * 1e30
    try:
        cases.run_cg_impl(hip, dims, flags, A, rhs, 2, 1e-3, 200)
    except RuntimeError as e:
        assert "diverged" in str(e)


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("liquid", [False, True])
def test_solve_pressure_vs_reference(hip_backend, dims, liquid):
    flags, vel, phi = cases.pressure_inputs(dims, 6, liquid)
    a = cases.run_solve_pressure_pkg(dims, flags, vel, phi)
    if util.have_ref():
        b = cases.run_solve_pressure_ref(dims, flags, vel, phi)
    else:
        from mantaflow_amd import _lib
        _lib.use_library(util.build_oracle(), "cpu")
        b = cases.run_solve_pressure_pkg(dims, flags, vel, phi)
        _lib.reset()
    assert_bitexact(a["rhs"], b["rhs"], "rhs")
    _close(a["pressure"], b["pressure"], "pressure")
    _close(a["vel"], b["vel"], "vel")


def test_solve_pressure_plain_system_paths(hip, hip_backend):
    """solvePressure on a plain system takes mf_solve_pressure_fused (matrix-free set-up) when the rows are a multiple of 8 cells and the
    "rows" sweeps are on; with mf_set_mic_mode("levels") the library declines and the plugin falls back to mf_make_rhs +
    mf_make_laplace_matrix + mf_cg_solve.  Both against the oracle: rhs bit-exact, same iteration count, fields within 1e-5 -- and
    the two GPU paths agree bit for bit on rhs and in iteration count."""
    from mantaflow_amd import _lib, plugins
    dims = (32, 24, 16)
    flags, vel, _ = cases.pressure_inputs(dims, 21, False)
    a = cases.run_solve_pressure_pkg(dims, flags, vel, None)
    it_a = plugins.lastCgStats()["iterations"]
    assert hip.lib.cdll.mf_set_mic_mode(b"levels") == 0
    try:
        c = cases.run_solve_pressure_pkg(dims, flags, vel, None)
        it_c = plugins.lastCgStats()["iterations"]
    finally:
        assert hip.lib.cdll.mf_set_mic_mode(None) == 0
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_solve_pressure_pkg(dims, flags, vel, None)
    it_b = plugins.lastCgStats()["iterations"]
    _lib.reset()
    assert it_a == it_b == it_c and it_b > 3, (it_a, it_b, it_c)
    for got in (a, c):
        assert_bitexact(got["rhs"], b["rhs"], "rhs")
        _close(got["pressure"], b["pressure"], "pressure")
        _close(got["vel"], b["vel"], "vel")


@pytest.mark.parametrize("dims,box", [((64, 40, 24), (19, 45, 1, 22, 1, 23)), ((64, 40, 24), (1, 30, 1, 17, 1, 23)), ((64, 40, 24), (40, 63, 9, 31, 1, 23)),
                                      ((61, 40, 24), (21, 44, 1, 22, 1, 23))])
def test_solve_pressure_liquid_shortcuts(hip_backend, dims, box):
    """A body of liquid strictly inside the box, away from the x walls or against one of them: most 8 x 8 bundles of rows hold no fluid
    and the fluid keeps to a part of the x-range, so mf_cg_solve skips the empty bundles in its streaming kernels and trims the MIC sweeps
    to the fluid's x-range (chunks of 8 cells; 61 cells per row: on the padded internal system).  Against the oracle, which does neither:
    rhs bit-exact, the same iteration count, fields within 1e-5 -- with and without the ghost-fluid boundary (phi)."""
    from mantaflow_amd import _lib, plugins
    sx, sy, sz = dims
    x0, x1, y0, y1, z0, z1 = box
    flags = np.full((sz, sy, sx), 4, np.int32)                 # empty
    flags[:, :, 0] = flags[:, :, -1] = flags[:, 0, :] = flags[:, -1, :] = 2
    flags[0] = flags[-1] = 2
    flags[z0:z1, y0:y1, x0:x1] = 1                              # fluid
    flags[3:7, 3:9, x0 + 2:x0 + 6] = 2                          # an obstacle inside the liquid
    vel = util.rand_vel(sx, sy, sz, 31, 0.5)
    zz, yy, xx = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
    inside = (xx >= x0) & (xx < x1) & (yy >= y0) & (yy < y1) & (zz >= z0) & (zz < z1)
    phi = np.where(inside, -0.6, 0.7).astype(np.float32) + (0.05 * np.sin(xx * 0.9 + yy * 0.4)).astype(np.float32)
    import ctypes
    for ph in (None, phi):
        a = cases.run_solve_pressure_pkg(dims, flags, vel, ph)
        it_a = plugins.lastCgStats()["iterations"]
        # the shortcut was taken: bundles skipped, sweeps trimmed to the chunks of 8 cells around [x0, x1] (one cell more behind the last
        # fluid cell: its Ai bit couples it)
        sc = (ctypes.c_int32 * 3)()
        assert _lib.get().cdll.mf_cg_last_shortcut(sc) == 0
        assert sc[0] == 1 and sc[1] == (x0 // 8) * 8 and sc[1] + sc[2] == min(((x1 + 1 + 7) // 8) * 8, ((sx + 7) // 8) * 8), list(sc)
        _lib.use_library(util.build_oracle(), "cpu")
        b = cases.run_solve_pressure_pkg(dims, flags, vel, ph)
        it_b = plugins.lastCgStats()["iterations"]
        _lib.reset()
        assert it_a == it_b and it_b > 3, (it_a, it_b)
        assert_bitexact(a["rhs"], b["rhs"], "rhs")
        _close(a["pressure"], b["pressure"], "pressure")
        _close(a["vel"], b["vel"], "vel")
        # outside the liquid the pressure is exactly what the reference leaves there
        assert_bitexact(np.where(flags == 1, 0, a["pressure"]).astype(np.float32), np.where(flags == 1, 0, b["pressure"]).astype(np.float32),
                        "pressure outside the fluid")


@pytest.mark.parametrize("terms,liquid", cases.PRESSURE_OPTIONAL_CASES)
@pytest.mark.parametrize("dims", [(24, 20, 16), cases.SIZE_2D])
def test_solve_pressure_optional_terms(hip_backend, dims, terms, liquid):
    """solvePressure with perCellCorr, fractions, obvel, curv + surfTens on the GPU against the compiled reference (the oracle
    where the reference .so did not travel): rhs bit-exact, identical CG path, fields within 1e-5"""
    flags, vel, phi = cases.pressure_inputs(dims, 12, liquid)
    extra = cases.pressure_optional_terms(dims, 13, terms)
    kw = dict(surfTens=0.7) if "curv" in terms else {}
    a = cases.run_solve_pressure_pkg(dims, flags, vel, phi, extra=extra, **kw)
    if util.have_ref():
        b = cases.run_solve_pressure_ref(dims, flags, vel, phi, extra=extra, **kw)
    else:
        from mantaflow_amd import _lib
        _lib.use_library(util.build_oracle(), "cpu")
        b = cases.run_solve_pressure_pkg(dims, flags, vel, phi, extra=extra, **kw)
        _lib.reset()
    assert_bitexact(a["rhs"], b["rhs"], "rhs")
    _close(a["pressure"], b["pressure"], "pressure")
    _close(a["vel"], b["vel"], "vel")


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D, (33, 17, 9), (16, 12, 10), (40, 17, 12)])
@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("order,clampMode", [(1, 2), (2, 1), (2, 2)])
@pytest.mark.parametrize("orderTrace", [1, 2])
def test_advect(hip_backend, oracle, dims, kind, order, clampMode, orderTrace):
    from mantaflow_amd import _lib
    sx, sy, sz = dims
    flags, vel = cases.advect_inputs(dims, 9, vmax=2.5, outflow=(kind == 2))
    field = util.rand_real((sz, sy, sx), 10) if kind == 0 else util.rand_vel(sx, sy, sz, 10)
    kw = dict(order=order, clampMode=clampMode, orderTrace=orderTrace, strength=0.8 if order == 2 else 1.0)
    a = cases.run_advect_pkg(dims, 0.9, flags, vel, field, kind, **kw)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_advect_pkg(dims, 0.9, flags, vel, field, kind, **kw)
    _lib.reset()
    assert_bitexact(a, b, "advectSemiLagrange kind=%d" % kind)


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D, (40, 17, 12)])
@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("order,orderTrace", [(1, 1), (2, 1), (2, 2)])
def test_advect_cubic(hip_backend, oracle, dims, kind, order, orderTrace):
    """orderSpace=2 (cubic interpolation, util/interpolHigh.h): bit-exact against the oracle (pinned to the compiled reference)"""
    from mantaflow_amd import _lib
    sx, sy, sz = dims
    flags, vel = cases.advect_inputs(dims, 9, vmax=2.5, outflow=(kind == 2))
    field = util.rand_real((sz, sy, sx), 10) if kind == 0 else util.rand_vel(sx, sy, sz, 10)
    kw = dict(order=order, clampMode=2, orderTrace=orderTrace, orderSpace=2, strength=0.8 if order == 2 else 1.0)
    a = cases.run_advect_pkg(dims, 0.9, flags, vel, field, kind, **kw)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_advect_pkg(dims, 0.9, flags, vel, field, kind, **kw)
    _lib.reset()
    assert_bitexact(a, b, "advectSemiLagrange(orderSpace=2) kind=%d" % kind)


@pytest.mark.parametrize("dims", [(12, 10, 9), cases.SIZE_2D])
@pytest.mark.parametrize("with_ptype", [False, True])
def test_flip_transfers(hip_backend, dims, with_ptype):
    from mantaflow_amd import _lib
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 14, empty_top=True)
    vel, velOld = util.rand_vel(sx, sy, sz, 15), util.rand_vel(sx, sy, sz, 16)
    pos, pflag, pvel = util.make_particles(flags, 3, 17)
    ptype = None
    if with_ptype:
        ptype = (np.random.default_rng(18).integers(0, 4, pos.shape[1]) * 2).astype(np.int32)
    a = cases.run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel, ptype, 4 if with_ptype else 0, deterministic=True)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel, ptype, 4 if with_ptype else 0)
    _lib.reset()
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(12, 10, 9), cases.SIZE_2D])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("deleteInObstacle,stopInObstacle", [(False, True), (True, True), (False, False), (True, False)])
def test_advect_in_grid(hip_backend, dims, mode, deleteInObstacle, stopInObstacle):
    from mantaflow_amd import _lib
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 19, empty_top=True)
    vel = util.smooth_vel(sx, sy, sz, 20, 2.0)
    pos, pflag, _ = util.make_particles(flags, 2, 21)
    a = cases.run_advect_parts_pkg(dims, 0.8, flags, vel, pos, pflag, mode, deleteInObstacle, stopInObstacle)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_advect_parts_pkg(dims, 0.8, flags, vel, pos, pflag, mode, deleteInObstacle, stopInObstacle)
    _lib.reset()
    assert_bitexact(a[1], b[1], "particle flags")
    assert_bitexact(a[0], b[0], "particle positions")


def test_empty_particle_system(hip_backend):
    dims = (12, 10, 9)
    flags = util.make_flags(*dims, 14)
    pos, pflag, pvel = np.zeros((3, 0), np.float32), np.zeros(0, np.int32), np.zeros((3, 0), np.float32)
    r = cases.run_flip_pkg(dims, flags, util.rand_vel(*dims, 1), util.rand_vel(*dims, 2), pos, pflag, pvel)
    assert not r["p2g_vel"].any() and not r["p2g_weight"].any()


@pytest.mark.parametrize("dims", [(12, 10, 9), cases.SIZE_2D])
def test_glue(hip_backend, dims):
    from mantaflow_amd import _lib
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 26, empty_top=True)
    flags[flags.shape[0] // 2, 3, 3] |= util.STICK
    vel, density = util.rand_vel(sx, sy, sz, 27), util.rand_real((sz, sy, sx), 28)
    a = cases.run_glue_pkg(dims, 0.7, flags, vel, density)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_glue_pkg(dims, 0.7, flags, vel, density)
    _lib.reset()
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D, (33, 18, 9)])
def test_flip_glue(hip_backend, dims):
    from mantaflow_amd import _lib
    fl0, pos, pflag, pvel, vel = cases.flipglue_inputs(dims, 31)
    phi = util.rand_real((dims[2], dims[1], dims[0]), 35)
    for ph in (None, phi):
        a = cases.run_flipglue_pkg(dims, fl0, pos, pflag, pvel, vel, ph)
        _lib.use_library(util.build_oracle(), "cpu")
        b = cases.run_flipglue_pkg(dims, fl0, pos, pflag, pvel, vel, ph)
        _lib.reset()
        for k in b:
            assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D, (33, 18, 9), (64, 48, 40)])
def test_surface_pieces(hip_backend, dims):
    """benchmark_dam.py's free-surface / particle maintenance calls: HIP bit-identical to the oracle (which is pinned to
    the compiled reference by tests/test_oracle_vs_reference.py::test_surface_pieces)"""
    from mantaflow_amd import _lib
    I = cases.surface_inputs(dims, 51)
    a = cases.run_surface_pkg(dims, I)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_surface_pkg(dims, I)
    _lib.reset()
    assert b["gpi_sys"].size > 100
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims,small", [((20, 14, 12), (10, 7, 6)), ((24, 30, 1), (12, 15, 1)), ((17, 11, 9), (9, 6, 5)), ((64, 48, 40), (32, 24, 20))])
def test_wavelet_turbulence_pieces(hip_backend, dims, small):
    """computeEnergy / computeWaveletCoeffs / vorticityConfinement / applyNoiseVec3 (scenes/waveletTurbulence.py): HIP bit-identical
    to the oracle, itself pinned to the compiled reference by test_oracle_vs_reference.py::test_wavelet_turbulence_pieces"""
    from mantaflow_amd import _lib
    inp = cases.turb_inputs(dims, small, 91)
    a = cases.run_turb_pkg(dims, *inp, small)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_turb_pkg(dims, *inp, small)
    _lib.reset()
    assert np.abs(b["noise_plain"] - inp[1]).max() > 1e-3
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims,per_cell,border", [((12, 10, 9), 3, False), (cases.SIZE_2D, 3, True), ((33, 18, 9), 4, True), ((64, 48, 40), 8, True), ((40, 33, 27), 27, False)])
@pytest.mark.parametrize("with_ptype", [False, True])
def test_apic_transfers(hip_backend, dims, per_cell, border, with_ptype):
    """apicMapPartsToMAC (ordered gather = the reference's serial scatter) / apicMapMACGridToParts: HIP bit-identical to the
    oracle, itself pinned to the compiled reference (test_oracle_vs_reference.py::test_apic_transfers); `border` adds
    particles in wall cells, whose face stencils wrap to the next grid row exactly as the reference's flat index does"""
    from mantaflow_amd import _lib
    flags, vel, pos, pflag, pvel, cp = cases.apic_inputs(dims, 61, per_cell, include_border=border)
    ptype = (np.random.default_rng(18).integers(0, 4, pos.shape[1]) * 2).astype(np.int32) if with_ptype else None
    ex = 4 if with_ptype else 0
    a = cases.run_apic_pkg(dims, flags, vel, pos, pflag, pvel, cp, ptype, ex)
    a2 = cases.run_apic_pkg(dims, flags, vel, pos, pflag, pvel, cp, ptype, ex, with_mass=False)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_apic_pkg(dims, flags, vel, pos, pflag, pvel, cp, ptype, ex)
    _lib.reset()
    for k in b:
        assert_bitexact(a[k], b[k], k)
    assert_bitexact(a2["apic_vel"], b["apic_vel"], "apic_vel without a mass grid")


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D, (33, 18, 9), (64, 48, 40)])
def test_cg_solve_diffusion(hip_backend, dims):
    """cgSolveDiffusion: same iteration counts as the oracle, fields within 1e-5 (fp64 partial sums combined in another order)"""
    from mantaflow_amd import _lib
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 81, obstacles=True)
    real, vel = util.rand_real((sz, sy, sx), 82), util.rand_vel(sx, sy, sz, 83)
    a = cases.run_diffusion_pkg(dims, flags, real, vel)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_diffusion_pkg(dims, flags, real, vel)
    _lib.reset()
    assert (a["iters"] == b["iters"]).all(), (a["iters"], b["iters"])
    _close(a["real"], b["real"], "diffused Real grid")
    _close(a["mac"], b["mac"], "diffused MAC grid")


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D])
def test_reset_outflow(hip_backend, dims):
    from mantaflow_amd import _lib
    inp = cases.outflow_inputs(dims, 71)
    a = cases.run_outflow_pkg(dims, *inp)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_outflow_pkg(dims, *inp)
    _lib.reset()
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(13, 11, 9), (16, 12, 1), (40, 33, 27)])
def test_shape_levelsets(hip, oracle, dims):
    """mf_shape_levelset (Box / Sphere / Cylinder signed distance fields): HIP bit-identical to the oracle, which
    test_oracle_vs_reference.py::test_shape_levelsets pins to the compiled reference"""
    sx, sy, sz = dims
    shapes = [(0, [1.5, 2.25, 0.5, sx * 0.6, sy * 0.7, max(sz * 0.8, 1.0), 0, 0, 0, 0, 0, 0]),
              (0, [-2.0, 1.0, -1.0, sx + 3.0, sy * 0.4, sz + 2.0, 0, 0, 0, 0, 0, 0]),
              (1, [sx * 0.5, sy * 0.4, sz * 0.5, sx * 0.27, 1.0, 1.0, 1.0, 0, 0, 0, 0, 0]),
              (1, [sx * 0.3, sy * 0.6, sz * 0.5, sx * 0.2, 1.5, 0.75, 2.0, 0, 0, 0, 0, 0]),
              (2, [sx * 0.5, sy * 0.1, sz * 0.5, sx * 0.14, 0.0, 1.0, 0.0, sy * 0.02, 0, 0, 0, 0]),
              (2, [sx * 0.3, sy * 0.2, sz * 0.5, sx * 0.08, 0.6, 0.64, 0.48, sx * 0.15, 0, 0, 0, 0])]
    for kind, q in shapes:
        qa = (ctypes.c_float * 12)(*[float(np.float32(v)) for v in q])
        out = []
        for impl in (hip, oracle):
            phi = impl.dev(np.zeros((sz, sy, sx), np.float32))
            impl.call("mf_shape_levelset", sx, sy, sz, kind, qa, phi, None)
            impl.sync()
            out.append(impl.host(phi))
        assert np.isfinite(out[1]).mean() > 0.99
        assert_bitexact(out[0], out[1], "shape kind %d" % kind)
        # Shape.applyToGrid on the four grid kinds, with obstacle cells respected
        flags = util.make_flags(sx, sy, sz, 88, obstacles=True)
        val = (ctypes.c_float * 3)(1.5, -2.25, 3.0)
        for gk in (0, 1, 2, 3):
            shp = (sz, sy, sx) if gk in (0, 3) else (3, sz, sy, sx)
            base = np.full(shp, 7, np.int32) if gk == 3 else util.rand_real(shp, 89).astype(np.float32)
            res = []
            for impl in (hip, oracle):
                g = impl.dev(base.copy())
                impl.call("mf_shape_apply_to_grid", sx, sy, sz, kind, qa, gk, g, val, impl.dev(flags), None)
                impl.sync()
                res.append(impl.host(g))
            assert_bitexact(res[0], res[1], "applyToGrid shape %d grid kind %d" % (kind, gk))


def test_dam_break_steps_match_oracle(hip_backend):
    """three steps of a ghost-fluid FLIP dam break (benchmark_dam.py's loop) on the GPU = the same steps on the oracle:
    flags and particle types bit-exact, CG iteration counts identical, fields within 1e-5 (deterministic P2G)"""
    from mantaflow_amd import _lib
    a = cases.run_dam_pkg(24, 3)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_dam_pkg(24, 3)
    _lib.reset()
    assert a["iters"] == b["iters"] and min(b["iters"]) > 3
    assert b["pos"].shape[1] > 10000 and (b["flags"] & 1).sum() > 1000
    assert_bitexact(a["flags"], b["flags"], "flags")
    assert_bitexact(a["ptype"], b["ptype"], "ptype")
    for k in ("pos", "pvel", "vel", "pres", "phi"):
        _close(a[k], b[k], k)


@pytest.mark.parametrize("dims,per_cell", [((64, 48, 40), 8), ((96, 80, 1), 4), ((40, 33, 27), 27)])
def test_ordered_p2g_bitexact_at_scale(hip_backend, dims, per_cell):
    """the default (ordered) particle->grid transfer: hundreds of thousands of shuffled particles, up to 27 per cell,
    ptype exclusion and deleted particles -- bit-identical to the oracle's serial scatter, and to itself on a re-run"""
    from mantaflow_amd import _lib
    flags = util.make_flags(*dims, 61, empty_top=True)
    vel, velOld = util.rand_vel(*dims, 62), util.rand_vel(*dims, 63)
    pos, pflag, pvel = util.make_particles(flags, per_cell, 64)
    rng = np.random.default_rng(65)
    perm = rng.permutation(pos.shape[1])             # particles are NOT cell-ordered
    pos, pflag, pvel = np.ascontiguousarray(pos[:, perm]), np.ascontiguousarray(pflag[perm]), np.ascontiguousarray(pvel[:, perm])
    ptype = rng.choice(np.array([1, 4, 1], np.int32), pos.shape[1]).astype(np.int32)
    keys = ("p2g_vel", "p2g_velOld", "p2g_weight", "p2g_real", "p2g_vec3")
    a = cases.run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel, ptype=ptype, exclude=4)
    a2 = cases.run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel, ptype=ptype, exclude=4)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel, ptype=ptype, exclude=4)
    _lib.reset()
    assert pos.shape[1] > 15000
    for k in keys:
        assert_bitexact(a[k], b[k], k)
        assert_bitexact(a[k], a2[k], k + " (re-run)")


def test_ordered_p2g_crowded_cells(hip_backend):
    """the binning pass of the ordered transfer with crowded cells: a cell with 6000 particles (beyond the workgroup sort's LDS:
    rank counting), cells with hundreds (workgroup bitonic sort) and the usual few per cell (per-cell insertion sort), particle
    order shuffled -- bit-identical to the oracle's serial scatter"""
    from mantaflow_amd import _lib
    dims = (20, 16, 12)
    flags = util.make_flags(*dims, 71, empty_top=False)
    vel, velOld = util.rand_vel(*dims, 72), util.rand_vel(*dims, 73)
    pos, pflag, pvel = util.make_particles(flags, 3, 74)
    rng = np.random.default_rng(75)
    crowds = [((7, 6, 5), 6000), ((8, 6, 5), 700), ((7, 7, 5), 300), ((12, 3, 9), 40), ((1, 1, 1), 5000)]
    extra = [np.array(c, np.float32)[:, None] + rng.uniform(0.0, 1.0, (3, m)).astype(np.float32) for c, m in crowds]
    pos = np.concatenate([pos] + extra, axis=1)
    n = pos.shape[1]
    perm = rng.permutation(n)
    pos = np.ascontiguousarray(pos[:, perm])
    pflag = np.concatenate([pflag, np.zeros(n - pflag.shape[0], np.int32)])[perm]
    pvel = np.ascontiguousarray(np.concatenate([pvel, rng.normal(0, 0.5, (3, n - pvel.shape[1])).astype(np.float32)], axis=1)[:, perm])
    keys = ("p2g_vel", "p2g_velOld", "p2g_weight", "p2g_real", "p2g_vec3")
    a = cases.run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel)
    _lib.reset()
    for k in keys:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("orderSpace", [1, 2])
@pytest.mark.parametrize("case", range(len(cases.INTERP_CASES)))
def test_interpolate_between_grid_sizes(hip_backend, case, orderSpace):
    from mantaflow_amd import _lib
    sd, td, scale, offset, size = cases.INTERP_CASES[case]
    fields = {"real": util.rand_real((sd[2], sd[1], sd[0]), 71), "vec": util.rand_vel(*sd, 72)}
    a = cases.run_interp_pkg(sd, td, scale, offset, size, fields, orderSpace)
    _lib.use_library(util.build_oracle(), "cpu")
    b = cases.run_interp_pkg(sd, td, scale, offset, size, fields, orderSpace)
    _lib.reset()
    for k in b:
        assert_bitexact(a[k], b[k], "%s case %d orderSpace %d" % (k, case, orderSpace))


def test_reductions_and_elementwise(hip, oracle):
    n = 1 << 20
    a, b = util.rand_real((n + 3,), 30, 3.0), util.rand_real((n + 3,), 31, 2.0)
    for impl_pair in [(hip, oracle)]:
        res = []
        for impl in impl_pair:
            da, db = impl.dev(a), impl.dev(b)
            d = ctypes.c_double(); f = ctypes.c_float(); lo = ctypes.c_float(); hi = ctypes.c_float()
            impl.call("mf_grid_dot", n + 3, da, db, ctypes.byref(d), None)
            impl.call("mf_grid_max_abs", n + 3, da, ctypes.byref(f), None)
            impl.call("mf_grid_min_max", n + 3, da, ctypes.byref(lo), ctypes.byref(hi), None)
            impl.call("mf_grid_scaled_add", n + 3, da, db, 0.37, None)
            impl.call("mf_update_search_vec", n + 3, db, da, -1.7, None)
            impl.call("mf_grid_clamp", n + 3, da, -1.0, 1.5, None)
            impl.call("mf_grid_safe_divide", n + 3, db, da, None)
            impl.sync()
            res.append((d.value, f.value, lo.value, hi.value, impl.host(da), impl.host(db)))
        h, o = res
        assert abs(h[0] - o[0]) <= 1e-12 * abs(o[0]) and np.float32(h[0]) == np.float32(o[0])
        assert h[1:4] == o[1:4]
        assert_bitexact(h[4], o[4], "axpy/clamp")
        assert_bitexact(h[5], o[5], "xpay/safeDivide")


# ---- BASELINE.json size (256^3): direct comparison of the stencil + preconditioner and whole-solve properties ----
def test_256_apply_matrix_and_mic_vs_oracle(hip, oracle):
    dims = (256, 256, 256)
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 41, obstacles=True, empty_top=True)
    A = cases.run_laplace_impl(hip, dims, flags, None)
    Ao = cases.run_laplace_impl(oracle, dims, flags, None)
    for x, y in zip(A, Ao):
        assert_bitexact(x, y, "MakeLaplaceMatrix 256^3")
    src = util.rand_real((sz, sy, sx), 42)
    assert_bitexact(cases.run_apply_matrix_impl(hip, dims, flags, A, src), cases.run_apply_matrix_impl(oracle, dims, flags, A, src), "ApplyMatrix 256^3")
    ap, dst = cases.run_mic_impl(hip, dims, flags, A, src)
    ap_o, dst_o = cases.run_mic_impl(oracle, dims, flags, A, src)
    assert_bitexact(ap, ap_o, "Aprecond 256^3")
    assert_bitexact(dst, dst_o, "MIC apply 256^3")


def test_256_smoke_step_properties(hip_backend):
    """one full 256^3 smoke step (advect density + velocity with MacCormack, setWallBcs, solvePressure):
    divergence-free result, residual below cgAccuracy, constants advect to constants, run-to-run determinism."""
    from mantaflow_amd import core, plugins
    dims = (256, 256, 256)
    sx, sy, sz = dims
    s = cases._mk_solver(dims, 1.0)
    fl = core.FlagGrid(s); fl.initDomain(); fl.fillGrid()
    vel, dens, pres = core.MACGrid(s), core.Grid(s), core.Grid(s)
    v0 = util.smooth_vel(sx, sy, sz, 43, 2.0)
    cases.soa_to_grid(vel, v0)
    plugins.setWallBcs(fl, vel)
    dens.setConst(0.75)
    plugins.advectSemiLagrange(fl, vel, dens, order=2)
    d = cases.grid_to_soa(dens)
    inner = d[1:-1, 1:-1, 1:-1]
    assert np.all(inner == np.float32(0.75)), "a constant field must advect to the same constant (interior)"
    assert not d[0].any() and not d[:, 0].any() and not d[:, :, 0].any(), "bnd=1 kernels leave the border of a fresh grid at 0"
    plugins.advectSemiLagrange(fl, vel, vel, order=2)
    plugins.setWallBcs(fl, vel)
    vin = cases.grid_to_soa(vel).copy()
    rhs = core.Grid(s)
    plugins.solvePressure(vel, pres, fl, retRhs=rhs)
    st = plugins.lastCgStats()
    assert 1 <= st["iterations"] < 384 and st["residual"] < 1e-3, st
    # divergence of the projected field (MakeRhs again) must be below the solver's max-norm tolerance scale
    rhs2 = core.Grid(s)
    plugins.computePressureRhs(rhs2, vel, pres, fl)
    div_before, div_after = rhs.getMaxAbs(), rhs2.getMaxAbs()
    assert div_after < 5e-3 and div_after < 1e-2 * div_before, (div_before, div_after)
    # determinism: same inputs -> bit-identical pressure
    p1 = cases.grid_to_soa(pres).copy()
    cases.soa_to_grid(vel, vin)
    plugins.solvePressure(vel, pres, fl)
    assert plugins.lastCgStats()["iterations"] == st["iterations"]
    assert np.array_equal(p1, cases.grid_to_soa(pres)), "solvePressure must be run-to-run deterministic"


def test_128_flip_round_trip(hip_backend):
    """128^3 FLIP transfers (config 3 shape): P2G of a constant particle velocity gives that constant wherever a
    weight landed; G2P of it returns it; advectInGrid keeps particles inside the domain."""
    from mantaflow_amd import core, plugins
    dims = (128, 128, 128)
    sx, sy, sz = dims
    s = cases._mk_solver(dims, 0.5)
    fl = core.FlagGrid(s); fl.initDomain(); fl.fillGrid()
    flags = cases.grid_to_soa(fl)
    sub = flags.copy(); sub[:, 76:, :] = 2; sub[:, :, 52:] = 2     # fluid block 0.4 x 0.6 x 1.0 of the domain
    pos, pflag, _ = util.make_particles(np.where(sub == 1, 1, 2).astype(np.int32), 8, 44, include_border=False, deleted_frac=0.0)
    pp = cases._mk_parts(s, pos, pflag)
    const = np.tile(np.array([[0.3], [-0.2], [0.1]], np.float32), (1, pp.np))
    pv = cases._pd_vec3(s, pp, const)
    v, vo, w = core.MACGrid(s), core.MACGrid(s), core.VecGrid(s)
    plugins.mapPartsToMAC(fl, v, vo, pp, pv, w)
    g, wt = cases.grid_to_soa(v), cases.grid_to_soa(w)
    for c, val in enumerate((0.3, -0.2, 0.1)):
        m = wt[c] > 0
        assert m.sum() > 1000
        assert np.allclose(g[c][m], val, rtol=2e-5, atol=0), "P2G of a constant must return the constant"
    pv2 = cases._pd_vec3(s, pp, np.zeros_like(const))
    full = core.MACGrid(s); full.setConst(core.vec3(0.3, -0.2, 0.1))
    plugins.mapMACToParts(fl, full, pp, pv2)
    assert np.allclose(cases._pd_get(pv2, pp.np), const, rtol=1e-6)
    cases.soa_to_grid(v, util.smooth_vel(sx, sy, sz, 45, 3.0))
    pp.advectInGrid(fl, v, 2, deleteInObstacle=False)
    s.sync()
    p = pp.get_positions()
    assert p.min() >= 0 and (p < np.array([sx, sy, sz])).all()
    assert not (flags[p[:, 2].astype(int), p[:, 1].astype(int), p[:, 0].astype(int)] & 2).all()

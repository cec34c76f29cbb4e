"""Shared test plumbing: loading the three implementations of the ABI, seeded scene generators, comparisons.

  hip     mantaflow_amd/csrc/libmanta_hip.so      product (device pointers; needs a GPU to run)
  oracle  oracle/libmanta_oracle.so               plain-C restatement (host pointers)
  ref     oracle/_ref/libmanta_ref.so             the reference's own C++ behind oracle/ref_shim.cpp (host pointers);
                                                  built here by oracle/ref.mk, travels to the GPU box as a binary
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB = os.path.join(ROOT, "mantaflow_amd", "csrc", "libmanta_hip.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "libmanta_oracle.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libmanta_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

import sys
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from mantaflow_amd import _lib  # noqa: E402

c_i, c_l, c_f, c_d, c_p, c_s = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p, ctypes.c_char_p

# argtypes of the shim's ref_* entry points (oracle/ref_shim.cpp)
REF_PROTOS = {
    "ref_apply_matrix": [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "ref_make_laplace_matrix": [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    "ref_mic_init": [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    "ref_mic_apply": [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "ref_cg_solve": [c_i, c_i, c_i] + [c_p] * 11 + [c_i, c_f, c_i, c_i, c_p],
    "ref_compute_pressure_rhs": [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_f],
    "ref_solve_pressure": [c_i, c_i, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p, c_f, c_p],
    "ref_correct_velocity": [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_f],
    "ref_advect_semi_lagrange": [c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_i, c_i],
    "ref_map_parts_to_mac": [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_i],
    "ref_cg_solve_diffusion": [c_i, c_i, c_i, c_p, c_p, c_i, c_f, c_f, c_f],
    "ref_shape_apply": [c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p],
    "ref_reset_outflow": [c_i, c_i, c_i, c_p, c_p, c_p, c_l, c_l, c_p, c_p, c_p],
    "ref_apic_map_parts_to_mac": [c_i, c_i, c_i, c_p, c_p, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i],
    "ref_apic_map_mac_to_parts": [c_i, c_i, c_i, c_p, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i],
    "ref_map_mac_to_parts": [c_i, c_i, c_i, c_p, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_i],
    "ref_flip_velocity_update": [c_i, c_i, c_i, c_p, c_p, c_p, c_l, c_l, c_p, c_p, c_p, c_f, c_p, c_i],
    "ref_map_parts_to_grid": [c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_l, c_p, c_p, c_p],
    "ref_map_grid_to_parts": [c_i, c_i, c_i, c_i, c_p, c_l, c_l, c_p, c_p, c_p],
    "ref_advect_in_grid": [c_i, c_i, c_i, c_p, c_p, c_l, c_l, c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_p, c_i],
    "ref_set_wall_bcs": [c_i, c_i, c_i, c_p, c_p, c_p],
    "ref_add_buoyancy": [c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_i],
    "ref_add_gravity": [c_i, c_i, c_i, c_f, c_p, c_p, c_f, c_f, c_f, c_p, c_i],
    "ref_grid_max_abs": [c_l, c_p, c_p],
    "ref_grid_sum_sqr": [c_l, c_p, c_p],
    "ref_extrapolate_mac_simple": [c_i, c_i, c_i, c_p, c_p, c_i, c_i],
    "ref_extrapolate_mac_from_weight": [c_i, c_i, c_i, c_p, c_p, c_i],
    "ref_mark_fluid_cells": [c_i, c_i, c_i, c_p, c_l, c_l, c_p, c_p, c_p, c_i, c_p],
    "ref_project_out_of_bnd": [c_i, c_i, c_i, c_l, c_l, c_p, c_p, c_f, ctypes.c_char_p, c_p, c_i],
    "ref_push_out_of_obs": [c_i, c_i, c_i, c_l, c_l, c_p, c_p, c_p, c_f, c_f, c_p, c_i],
    "ref_grid_particle_index": [c_i, c_i, c_i, c_l, c_l, c_p, c_p, c_p, c_p, c_p],
    "ref_union_particle_levelset": [c_i, c_i, c_i, c_l, c_l, c_p, c_p, c_p, c_f, c_p, c_i],
    "ref_extrapolate_ls_simple": [c_i, c_i, c_i, c_p, c_i, c_i, c_i],
    "ref_set_part_type": [c_i, c_i, c_i, c_p, c_l, c_l, c_p, c_p, c_i, c_i, c_i],
    "ref_mark_isolated_fluid_cell": [c_i, c_i, c_i, c_p, c_i],
    "ref_add_force_pvel": [c_l, c_l, c_p, c_f, c_f, c_f, c_f, c_p, c_i],
    "ref_update_velocity_from_delta_pos": [c_l, c_l, c_p, c_p, c_p, c_f, c_p, c_i],
    "ref_euler_step": [c_l, c_l, c_p, c_p, c_f, c_p, c_i],
    "ref_levelset_join": [c_l, c_p, c_p],
    "ref_levelset_subtract": [c_l, c_p, c_p, c_p, c_i],
    "ref_grid_set_bound": [c_i, c_i, c_i, c_p, c_f, c_i],
    "ref_grid_file": [c_i, c_i, c_i, c_i, c_i, ctypes.c_char_p, c_p],
    "ref_compute_energy": [c_i, c_i, c_i, c_p, c_p, c_p],
    "ref_compute_wavelet_coeffs": [c_i, c_i, c_i, c_p],
    "ref_vorticity_confinement": [c_i, c_i, c_i, c_p, c_p, c_f, c_p],
    "ref_set_open_bound": [c_i, c_i, c_i, c_p, c_i, ctypes.c_char_p, c_i],
    "ref_apply_noise_vec3": [c_i, c_i, c_i, c_f, c_p, c_p, c_i, c_p, c_f, c_f, c_p, c_i, c_i, c_i, c_p],
    "ref_waveletturbulence": [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "ref_simpleplume": [c_i, c_i, c_i, c_p, c_p],
    "ref_shape_levelset": [c_i, c_i, c_i, c_i, c_p, c_p],
    "ref_noise_tile": [c_p],
    "ref_density_inflow": [c_i, c_i, c_i, c_f, c_p, c_p, c_i, c_p, c_i, c_p, c_f, c_f],
    "ref_interpolate_grid": [c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i],
    "ref_sample_flags_with_particles": [c_i, c_i, c_i, c_p, c_i, c_f, c_l, c_p, c_p],
    "ref_init_domain": [c_i, c_i, c_i, c_p, c_i, c_s, c_s, c_s, c_s, c_i],
}


def build_oracle():
    if not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(os.path.join(ROOT, "oracle", "manta_oracle.c")):
        subprocess.check_call(["make", "-s", "-f", os.path.join(ROOT, "oracle", "Makefile")])
    return ORACLE_LIB


def have_ref():
    return os.path.exists(REF_LIB)


_ref = None


def ref():
    """ctypes handle of the compiled reference (None if it was not built / did not travel)."""
    global _ref
    if _ref is None and have_ref():
        L = ctypes.CDLL(REF_LIB)
        for name, at in REF_PROTOS.items():
            fn = getattr(L, name)
            fn.argtypes, fn.restype = at, ctypes.c_int
        L.mf_last_error.restype = ctypes.c_char_p
        _ref = L
    return _ref


def P(a):
    """numpy array / torch tensor / None -> void*"""
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return ctypes.c_void_p(a.data_ptr())
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def refcall(name, *args):
    L = ref()
    rc = getattr(L, name)(*[P(a) if isinstance(a, (np.ndarray, torch.Tensor)) else a for a in args])
    if rc != 0:
        raise RuntimeError(L.mf_last_error().decode())


class Impl:
    """One implementation of the mf_* ABI + the device its arrays live on."""

    def __init__(self, which):
        if which == "oracle":
            self.lib = _lib.Library(build_oracle(), "cpu")
        elif which == "hip":
            self.lib = _lib.Library(HIP_LIB, "cuda")
        else:
            raise ValueError(which)
        self.device = self.lib.device
        self.which = which

    def dev(self, a):
        """numpy -> array usable by this implementation (torch tensor on its device)"""
        if a is None:
            return None
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def host(self, t):
        return t.detach().cpu().numpy()

    def call(self, name, *args):
        a2 = []
        for a in args:
            if isinstance(a, torch.Tensor):
                a2.append(ctypes.c_void_p(a.data_ptr()))
            elif isinstance(a, np.ndarray):
                raise TypeError("pass device arrays (Impl.dev) to Impl.call")
            else:
                a2.append(a)
        r = self.lib.call(name, *a2)
        return r

    def sync(self):
        if self.device != "cpu":
            torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------------------
# seeded scenes
# ---------------------------------------------------------------------------------------------------------
FLUID, OBS, EMPTY, INFLOW, OUTFLOW, OPEN, STICK = 1, 2, 4, 8, 16, 32, 64


def make_flags(sx, sy, sz, seed=0, obstacles=True, empty_top=False, outflow=False, bw=0):
    """initDomain(bw)+fillGrid pattern (grid.cpp:798-927) plus seeded obstacle blobs / an empty band / outflow cells."""
    rng = np.random.default_rng(seed)
    f = np.full((sz, sy, sx), FLUID, np.int32)
    w = bw + 1
    f[:, :, :w] = OBS; f[:, :, -w:] = OBS; f[:, :w, :] = OBS; f[:, -w:, :] = OBS
    if sz > 1:
        f[:w] = OBS; f[-w:] = OBS
    if obstacles:
        for _ in range(3):
            c = [rng.integers(2, max(3, s - 2)) for s in (sz, sy, sx)]
            r = rng.integers(1, max(2, min(sx, sy) // 6))
            zz, yy, xx = np.ogrid[:sz, :sy, :sx]
            m = (xx - c[2]) ** 2 + (yy - c[1]) ** 2 + ((zz - c[0]) ** 2 if sz > 1 else 0) <= r * r
            f[m & (f == FLUID)] = OBS
    if empty_top:
        band = f[:, (2 * sy) // 3:, :]
        band[band == FLUID] = EMPTY
    if outflow:
        col = f[:, :, -w - 2:-w]
        col[col != OBS] = EMPTY | OUTFLOW
    return f


def rand_real(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, shape) * scale).astype(np.float32)


def rand_vel(sx, sy, sz, seed, scale=1.0):
    """SoA MAC field [3][sz][sy][sx]; z component zero in 2-D"""
    v = rand_real((3, sz, sy, sx), seed, scale)
    if sz == 1:
        v[2] = 0
    return v


def smooth_vel(sx, sy, sz, seed, scale=1.0):
    """band-limited field: sum of a few sines, magnitude ~scale"""
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
    v = np.zeros((3, sz, sy, sx), np.float64)
    for c in range(3):
        for _ in range(3):
            k = rng.uniform(0.5, 3.0, 3) * 2 * np.pi / np.array([max(sz, 2), sy, sx])
            ph = rng.uniform(0, 2 * np.pi, 3)
            v[c] += rng.uniform(-1, 1) * np.sin(k[0] * zz + ph[0]) * np.sin(k[1] * yy + ph[1]) * np.sin(k[2] * xx + ph[2])
    v *= scale / max(np.abs(v).max(), 1e-9)
    if sz == 1:
        v[2] = 0
    return v.astype(np.float32)


def make_particles(flags, per_cell, seed, vel_scale=0.5, deleted_frac=0.02, include_border=True):
    """positions [3][np] (SoA), flags [np], velocities [3][np]: jittered sub-cell lattice inside fluid cells,
    plus a few particles in border / obstacle cells and a few flagged PDELETE."""
    rng = np.random.default_rng(seed)
    sz, sy, sx = flags.shape
    kk, jj, ii = np.nonzero(flags & FLUID)
    base = np.stack([ii, jj, kk], 0).astype(np.float32)
    pos = np.repeat(base, per_cell, axis=1) + rng.uniform(0.02, 0.98, (3, base.shape[1] * per_cell)).astype(np.float32)
    if sz == 1:
        pos[2] = 0.5
    if include_border:
        nb = max(4, pos.shape[1] // 50)
        extra = np.stack([rng.uniform(0.0, sx - 1e-3, nb), rng.uniform(0.0, sy - 1e-3, nb),
                          rng.uniform(0.0, sz - 1e-3, nb) if sz > 1 else np.full(nb, 0.5)], 0).astype(np.float32)
        pos = np.concatenate([pos, extra], axis=1)
    n = pos.shape[1]
    perm = rng.permutation(n)
    pos = np.ascontiguousarray(pos[:, perm])
    pflag = np.zeros(n, np.int32)
    pflag[rng.random(n) < deleted_frac] = 1 << 10
    pflag[rng.random(n) < 0.05] |= 1      # PNEW
    pvel = (rng.normal(0, vel_scale, (3, n))).astype(np.float32)
    if sz == 1:
        pvel[2] = 0
    return pos, pflag, pvel


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else (a.view(np.int64) if a.dtype == np.float64 else a)


def assert_bitexact(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = bits(a) == bits(b)
    # +0 / -0 are the same value for every consumer on this path
    if a.dtype.kind == "f":
        same |= (a == 0) & (b == 0)
    if not same.all():
        bad = np.argwhere(~same)
        i = tuple(bad[0])
        raise AssertionError("%s: %d of %d values differ; first at %s: %r vs %r (max abs diff %g)" % (
            what, len(bad), a.size, i, a[i], b[i], np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))

"""Generates the golden vectors under tests/golden/ from the REFERENCE ITSELF (oracle/_ref/libmanta_ref.so, the
reference's own C++ compiled by oracle/ref.mk) -- run in the build container:  python tests/golden/make_golden.py
Each .npz holds the seeded inputs' parameters and the reference's outputs; inputs are regenerated from the seeds by
tests/cases.py, so the files stay small.  The files are data (arrays), not reference source."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cases  # noqa: E402
import util  # noqa: E402


def main():
    assert util.have_ref(), "build the reference first: make -f oracle/ref.mk"
    out = {}
    # pressure system: 16^3 (SPD, empty band) and 20x13x11
    for tag, dims, seed in [("a", (16, 16, 16), 5), ("b", (20, 13, 11), 7)]:
        flags, A, src = cases.system_inputs(dims, seed)
        out["apply_%s" % tag] = cases.run_apply_matrix_ref(dims, flags, A, src)
        ap, dst = cases.run_mic_ref(dims, flags, A, src)
        out["micinit_%s" % tag], out["micapply_%s" % tag] = ap, dst
        rhs = cases.cg_rhs(dims, flags, seed)
        x, st = cases.run_cg_ref(dims, flags, A, rhs, 2, 1e-3, 60)
        out["cg_%s" % tag], out["cgstat_%s" % tag] = x, np.array(st, np.float64)
    for tag, dims, liquid in [("smoke", (16, 16, 16), False), ("liquid", (20, 13, 11), True), ("2d", (24, 18, 1), True)]:
        flags, vel, phi = cases.pressure_inputs(dims, 6, liquid)
        r = cases.run_solve_pressure_ref(dims, flags, vel, phi)
        for k in ("rhs", "pressure", "vel"):
            out["sp_%s_%s" % (tag, k)] = r[k]
    dims = (14, 12, 10)
    sx, sy, sz = dims
    for kind in (0, 1, 2):
        flags, vel = cases.advect_inputs(dims, 9, vmax=2.5, outflow=(kind == 2))
        field = util.rand_real((sz, sy, sx), 10) if kind == 0 else util.rand_vel(sx, sy, sz, 10)
        for order, cm in ((1, 2), (2, 1), (2, 2)):
            out["adv_k%d_o%d_c%d" % (kind, order, cm)] = cases.run_advect_ref(dims, 0.9, flags, vel, field, kind, order=order, clampMode=cm,
                                                                             strength=0.8 if order == 2 else 1.0)
    dims = (12, 10, 9)
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 14, empty_top=True)
    vel, velOld = util.rand_vel(sx, sy, sz, 15), util.rand_vel(sx, sy, sz, 16)
    pos, pflag, pvel = util.make_particles(flags, 3, 17)
    r = cases.run_flip_ref(dims, flags, vel, velOld, pos, pflag, pvel)
    for k, v in r.items():
        out["flip_" + k] = v
    flags = util.make_flags(sx, sy, sz, 19, empty_top=True)
    v2 = util.smooth_vel(sx, sy, sz, 20, 2.0)
    pos, pflag, _ = util.make_particles(flags, 2, 21)
    for mode in (0, 1, 2):
        p, f = cases.run_advect_parts_ref(dims, 0.8, flags, v2, pos, pflag, mode, False, True)
        out["padv_m%d_pos" % mode], out["padv_m%d_flag" % mode] = p, f
    p, f = cases.run_advect_parts_ref(dims, 0.8, flags, v2, pos, pflag, 2, True, True)
    out["padv_del_pos"], out["padv_del_flag"] = p, f
    # FLIP glue and the free-surface pieces of benchmark_dam.py (8f-3)
    gd = (14, 12, 10)
    fl0, pos, pflag, pvel, vel = cases.flipglue_inputs(gd, 31)
    for k, v in cases.run_flipglue_ref(gd, fl0, pos, pflag, pvel, vel, None).items():
        out["glue_" + k] = v
    for k, v in cases.run_surface_ref(gd, cases.surface_inputs(gd, 51)).items():
        out["surf_" + k] = v
    # BASELINE config 0: scenes/simpleplume.py, 5 steps at res 16, through the reference's own classes (ref_simpleplume)
    r = cases.run_simpleplume_ref(16, 5)
    out["plume_density"], out["plume_vel"] = r["density"], r["vel"]
    # wavelet-turbulence pieces (scenes/waveletTurbulence.py)
    td, ts = (20, 14, 12), (10, 7, 6)
    for k, v in cases.run_turb_ref(td, *cases.turb_inputs(td, ts, 91), ts).items():
        out["turb_" + k] = v
    # scenes/waveletTurbulence.py end to end (2D res 32 x 6 steps, 3D res 16 x 4 steps) through the reference's own classes
    for tag, dim, res, steps in (("wlt2d_", 2, 32, 6), ("wlt3d_", 3, 16, 4)):
        for k, v in cases.run_wavelet_scene_ref(res, dim, steps).items():
            out[tag + k] = v
    # APIC transfers (plugin/apic.cpp)
    ad = (12, 10, 9)
    out.update(cases.run_apic_ref(ad, *cases.apic_inputs(ad, 61)))
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote %d arrays, %.1f KiB" % (len(out), os.path.getsize(os.path.join(HERE, "reference_vectors.npz")) / 1024))


if __name__ == "__main__":
    main()
